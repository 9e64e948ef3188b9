#!/usr/bin/env python3
"""Per-basic-block opcode-class histogram of one kernel in a gfx950 assembly file (hipcc -S / -save-temps).

Classes follow profiles/r01_valu_rates (tools/valu_rates.hip): FAST = the VALU opcodes that two or more resident waves
co-issue at ~2.2 cycles per wave-instruction (v_and/or/xor/not/add_u32/sub_u32/mov/ashrrev/lshrrev/bitop3 with VGPR or
literal operands); SLOW = every other VALU opcode (~4.4 cycles however many waves are resident); SALU; MEM (VMEM/SMEM/LDS);
OTHER (s_waitcnt, s_nop, branches are listed separately).

Usage: isa_opclass.py FILE.s KERNEL_SUBSTRING [--blocks] [--loop]
  --loop    restrict to the blocks between the largest backward branch target and its branch (the playout loop)
"""
import argparse, collections, re, sys

FAST = {"v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_ashrrev_i32",
        "v_lshrrev_b32", "v_bitop3_b32", "v_xnor_b32"}


def classify(op, operands=""):
    if op.startswith("v_"):
        if op.startswith("v_cmp") or op.startswith("v_cmpx"):
            return "SLOW"
        base = op.replace("_e32", "").replace("_e64", "")
        if base in FAST:
            # an SGPR operand moves the fast opcodes into the slow class (valu_rates: 'v_and_b32 with sgpr')
            if re.search(r"\bs\d+\b|\bs\[\d+:\d+\]|\bvcc|\bexec", operands.split(",", 1)[1] if "," in operands else ""):
                return "SLOW"
            return "FAST"
        return "SLOW"
    if op.startswith("s_"):
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_sleep", "s_setprio", "s_sendmsg", "s_code_end")):
            return "OTHER"
        if op.startswith(("s_branch", "s_cbranch")):
            return "BRANCH"
        if op.startswith(("s_load", "s_buffer_load", "s_store", "s_memtime", "s_memrealtime", "s_dcache")):
            return "MEM"
        return "SALU"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_", "ds_")):
        return "MEM"
    return "OTHER"


def parse_kernel(path, name):
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[A-Za-z_]\w*:", l) and name in l.split(":")[0]:
            start = i
            break
    if start is None:
        raise SystemExit(f"kernel containing '{name}' not found")
    blocks = []   # (label, [ (op, operands) ])
    cur = ("entry", [])
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end") or l.strip().startswith(".section") and blocks:
            break
        m = re.match(r"^(\.LBB[\w]+):", l)
        if m:
            blocks.append(cur)
            cur = (m.group(1), [])
            continue
        s = l.strip()
        if not s or s.startswith((";", ".", "//")):
            continue
        s = s.split(";")[0].strip()
        if not s:
            continue
        parts = s.split(None, 1)
        cur[1].append((parts[0], parts[1] if len(parts) > 1 else ""))
    blocks.append(cur)
    return lines[start].split(":")[0], blocks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file"); ap.add_argument("kernel")
    ap.add_argument("--blocks", action="store_true"); ap.add_argument("--loop", action="store_true")
    ap.add_argument("--top", type=int, default=25)
    a = ap.parse_args()
    kname, blocks = parse_kernel(a.file, a.kernel)
    idx = {b[0]: i for i, b in enumerate(blocks)}
    lo, hi = 0, len(blocks) - 1
    if a.loop:
        best = None
        for i, (lab, ins) in enumerate(blocks):
            for op, opr in ins:
                if op.startswith(("s_cbranch", "s_branch")):
                    t = opr.strip()
                    if t in idx and idx[t] <= i:
                        size = sum(len(b[1]) for b in blocks[idx[t]:i + 1])
                        if best is None or size > best[0]:
                            best = (size, idx[t], i)
        if best:
            _, lo, hi = best
    tot = collections.Counter(); ops = collections.Counter(); per_block = []
    for lab, ins in blocks[lo:hi + 1]:
        c = collections.Counter()
        for op, opr in ins:
            k = classify(op, opr)
            c[k] += 1; tot[k] += 1
            if k in ("FAST", "SLOW"):
                ops[(k, op.replace("_e32", "").replace("_e64", "") + ("(sgpr)" if k == "SLOW" and op.replace("_e32", "").replace("_e64", "") in FAST else ""))] += 1
        per_block.append((lab, len(ins), dict(c)))
    print(f"kernel {kname}\nblocks {lo}..{hi} of {len(blocks)}{' (largest loop)' if a.loop else ''}")
    valu = tot["FAST"] + tot["SLOW"]
    print("totals:", dict(tot), f"VALU {valu}, fast share {tot['FAST'] / max(valu, 1):.3f}")
    print("top VALU opcodes:")
    for (k, op), n in ops.most_common(a.top):
        print(f"  {n:5d}  {k:4s}  {op}")
    if a.blocks:
        for lab, n, c in per_block:
            if n >= 8:
                print(f"  {lab:16s} {n:5d} {c}")


if __name__ == "__main__":
    main()
