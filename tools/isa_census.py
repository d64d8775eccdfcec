#!/usr/bin/env python3
"""Instruction census of a kernel from hipcc's assembly (`hipcc -S --cuda-device-only`): the kernel's instruction stream is cut at
labels, s_barrier and branches into segments, each segment's instructions are counted by class.  For k_corr the segments of the
q loop ARE the barrier-delimited phases of one 5000-point sub-transform (phase 1: loads + product + radix-10 + pass-1 stores;
pass 2: radix-25; pass 3: radix-20 + rotation + accumulate).
Usage: tools/isa_census.py build/isa/acq_kernels.s <kernel-symbol-substring> [--segments]
       tools/isa_census.py build/isa/acq_kernels.s --assert-dma-wait"""
import collections
import re
import sys

CLASSES = [
    ("pk_fma", r"v_pk_fma_f32"), ("pk_mul", r"v_pk_mul_f32"), ("pk_add", r"v_pk_add_f32"),
    ("v_fma/mul/add f32", r"v_(fma|mul|add|sub|mac|fmac)_f32"), ("v_mov/readlane/dpp", r"v_(mov_b32|mov_b64|readlane|writelane|readfirstlane|accvgpr|cndmask|perm)"),
    ("v_int/addr", r"v_(add_u32|add_co|addc|lshl|lshr|and|or|xor|mul_u32|mul_lo|mul_hi|mad|bfe|sub_u32|lshl_add|add3|ashr|cmp|min|max|cvt|xad)"),
    ("ds_read", r"ds_read"), ("ds_write", r"ds_write"), ("ds_other", r"ds_"),
    ("vmem_load", r"(buffer_load|global_load|flat_load|scratch_load)"), ("vmem_store", r"(buffer_store|global_store|flat_store|scratch_store|global_atomic)"),
    ("s_load", r"s_(load|buffer_load)"), ("s_waitcnt", r"s_waitcnt"), ("s_barrier", r"s_barrier"), ("s_branch", r"s_(cbranch|branch)"),
    ("s_other", r"s_"), ("v_other", r"v_"),
]


def classify(op):
    for name, pat in CLASSES:
        if re.match(pat, op):
            return name
    return "other"


def kernel_body(path, needle):
    lines = open(path).read().splitlines()
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w+:", l) and needle in l.split(":")[0]:
            start = i
            break
    if start is None:
        raise SystemExit("kernel %r not found" % needle)
    body = []
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        body.append(l)
    return lines[start].split(":")[0], body


def assert_dma_wait(path):
    """Every kernel that refreshes an LDS table by LDS-DMA (`buffer_load ... lds`, k_corr<..., FOLD>) must drain its vector-memory
    counter before the barrier that publishes the table to the other waves: the source spells the wait out (inline asm after
    phase 1, acq_kernels.hip), this checks it reached the ISA -- one `s_waitcnt vmcnt(0)` in an inline-asm block per phase-1 copy
    (phase 1 ends in `s_setprio 0`), each followed by the s_barrier with no vector-memory instruction in between."""
    lines = open(path).read().splitlines()
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    checked = 0
    for a in starts:
        body = []
        for l in lines[a + 1:]:
            if l.startswith(".Lfunc_end"):
                break
            body.append(l.strip())
        if not any(re.match(r"buffer_load\w* .*\blds$", t) for t in body):
            continue
        name = lines[a].split(":")[0]
        waits = [i for i, t in enumerate(body) if t == "s_waitcnt vmcnt(0)" and i > 0 and body[i - 1].startswith(";;#ASMSTART")]
        n_phase1 = sum(1 for t in body if t == "s_setprio 0")
        if not waits or len(waits) != n_phase1:
            raise SystemExit(f"{name}: {len(waits)} explicit vmcnt(0) waits for {n_phase1} phase-1 copies")
        for w in waits:
            ok = False
            for t in body[w + 1:w + 12]:
                op = t.split()[0] if t and not t.startswith((";", ".")) else ""
                if op == "s_barrier":
                    ok = True
                    break
                if re.match(r"(buffer_|global_|flat_|scratch_)", op):
                    break
            if not ok:
                raise SystemExit(f"{name}: the explicit vmcnt(0) wait at body line {w} is not followed by the barrier")
        checked += 1
        print(f"{name}: {len(waits)} LDS-DMA waits in front of their barriers: ok")
    if not checked:
        raise SystemExit("no kernel with LDS-DMA found")


def main():
    if "--assert-dma-wait" in sys.argv:
        return assert_dma_wait(sys.argv[1])
    path, needle = sys.argv[1], sys.argv[2]
    name, body = kernel_body(path, needle)
    segs = []  # (label, Counter)
    cur, label = collections.Counter(), "entry"
    for l in body:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            m = re.match(r"^(\.LBB\w+):", t)
            if m:
                segs.append((label, cur))
                cur, label = collections.Counter(), m.group(1)
            continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            segs.append((label, cur))
            cur, label = collections.Counter(), m.group(1)
            continue
        op = t.split()[0]
        c = classify(op)
        cur[c] += 1
        if c == "s_barrier":
            segs.append((label, cur))
            cur, label = collections.Counter(), label + "+bar"
    segs.append((label, cur))
    total = collections.Counter()
    for _, c in segs:
        total.update(c)
    cols = [n for n, _ in CLASSES] + ["other"]
    print("kernel", name)
    print("%-22s" % "segment" + "".join("%8s" % c[:8] for c in cols) + "   VALU  all")
    def row(lbl, c):
        valu = sum(c[k] for k in ("pk_fma", "pk_mul", "pk_add", "v_fma/mul/add f32", "v_mov/readlane/dpp", "v_int/addr", "v_other"))
        print("%-22s" % lbl[:22] + "".join("%8d" % c[k] for k in cols) + "  %5d %5d" % (valu, sum(c.values())))
    for lbl, c in segs:
        if sum(c.values()) >= (1 if "--segments" in sys.argv else 40):
            row(lbl, c)
    row("TOTAL (static)", total)


if __name__ == "__main__":
    main()
