#!/usr/bin/env python3
"""Instruction mix of a kernel's loops, from the disassembly of the code objects inside comet_amd/libcomet_hip.so (no GPU needed).
A loop = the span between a backward branch and its target. Per loop: instructions by class (MFMA, other VALU, SALU, LDS, global / buffer / scratch
memory, waits, barriers) — what DESIGN.md's per-pass budgets ("96 MFMAs per barrier", "4 VALU per accumulator") can be checked against.

    python tools/isa_mix.py 'flat_scan_qr_kernelILi0ELi128ELi6ELb0E' [--min-insts 200]
"""
import argparse
import re
import subprocess
import sys
import tempfile
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import kernel_resources as kr  # noqa: E402

LLVM = kr.LLVM


def disassemble(so):
    out = []
    for co in kr.code_objects(Path(so)):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            out.append(subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True).stdout)
    return out


def classify(op):
    if op.startswith(("v_mfma", "v_smfmac")): return "mfma"
    if op.startswith("v_accvgpr"): return "acc_mov"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_load_lds", "buffer_load") ) and "lds" in op: return "vmem_to_lds"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_"): return "salu"
    return "other"


def kernels(text):
    cur, body = None, []
    for line in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
        if m:
            if cur: yield cur, body
            cur, body = m.group(2), []
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*// ([0-9A-F]+):", line)
        if m and cur:
            body.append((int(m.group(3), 16), m.group(1), m.group(2), line))
    if cur: yield cur, body


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel", help="regex on the mangled kernel name")
    ap.add_argument("--so", default=str(kr.ROOT / "comet_amd" / "libcomet_hip.so"))
    ap.add_argument("--min-insts", type=int, default=100)
    a = ap.parse_args()
    for text in disassemble(a.so):
        for name, body in kernels(text):
            if not re.search(a.kernel, name):
                continue
            addr_index = {ad: i for i, (ad, *_rest) in enumerate(body)}
            print(f"== {name}: {len(body)} instructions")
            tot = {}
            for _ad, op, _args, _l in body:
                tot[classify(op)] = tot.get(classify(op), 0) + 1
            print("   whole kernel:", ", ".join(f"{k} {v}" for k, v in sorted(tot.items(), key=lambda t: -t[1])))
            loops = []
            for i, (ad, op, args, line) in enumerate(body):
                if not op.startswith(("s_cbranch", "s_branch")):
                    continue
                m = re.search(r"\+0x([0-9a-f]+)>", line)
                if not m:
                    continue
                tgt = body[0][0] + int(m.group(1), 16)
                if tgt <= ad and tgt in addr_index:
                    loops.append((addr_index[tgt], i))
            for lo, hi in sorted(loops, key=lambda t: t[0] - t[1]):
                n = hi - lo + 1
                if n < a.min_insts:
                    continue
                mix = {}
                for _ad, op, _args, _l in body[lo:hi + 1]:
                    mix[classify(op)] = mix.get(classify(op), 0) + 1
                inner = sum(1 for l2, h2 in loops if lo < l2 and h2 < hi)
                print(f"   loop +0x{body[lo][0] - body[0][0]:x} .. +0x{body[hi][0] - body[0][0]:x}: {n} instructions" + (f" (encloses {inner} inner loops)" if inner else "") + ": "
                      + ", ".join(f"{k} {v}" for k, v in sorted(mix.items(), key=lambda t: -t[1])))
    return 0


if __name__ == "__main__":
    sys.exit(main())
