#!/usr/bin/env python3
"""Static resources of every gfx950 kernel in comet_amd/libcomet_hip.so, read from the code objects' own metadata (no GPU needed):
registers (`v+a` = .vgpr_count: the wave's allocation in the unified file, architectural VGPRs + AGPRs), SGPRs, static LDS and scratch bytes, spills (VGPR spills
go to scratch memory, SGPR spills to VGPR lanes), workgroup size, and the waves per SIMD the register file allows (512 per SIMD lane on gfx950, granule 8; dynamic LDS is
set at launch and not in the metadata). What DESIGN.md claims about occupancy ("one wave per SIMD with 512 registers", "no spills") can be checked here.

    python tools/kernel_resources.py [--so comet_amd/libcomet_hip.so] [--all]  > profiles/rNN_kernel_resources.txt
"""
import argparse
import re
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LLVM = Path("/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(so: Path):
    """the gfx950 ELF images inside the library's .hip_fatbin section (one clang offload bundle per translation unit that holds kernels)"""
    with tempfile.TemporaryDirectory() as td:
        fat = Path(td) / "fat.bin"
        subprocess.run([str(LLVM / "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(so), str(fat)], check=True)
        d = fat.read_bytes()
    out = []
    for m in re.finditer(re.escape(MAGIC), d):
        base = m.start()
        (n,) = struct.unpack_from("<Q", d, base + len(MAGIC))
        p = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", d, p); p += 24
            triple = d[p:p + tl].decode(); p += tl
            if "gfx950" in triple and size:
                out.append(d[base + off: base + off + size])
    return out


def kernels_of(elf: bytes):
    import yaml
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf); f.flush()
        txt = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", f.name], check=True, capture_output=True, text=True).stdout
    ks = []
    for doc in re.findall(r"^\s*---\n(.*?)^\.\.\.", txt, flags=re.S | re.M):
        md = yaml.safe_load(doc) or {}
        ks += [{k.lstrip("."): v for k, v in rec.items() if k != ".args"} for rec in md.get("amdhsa.kernels", [])]
    return ks


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines() if r.returncode == 0 else names


def waves_per_simd(vgpr_total):
    """.vgpr_count is the wave's whole allocation in the unified register file (architectural VGPRs rounded up to 4 + AGPRs, gfx90a and later); granule 8, 512 per SIMD lane"""
    tot = max(8, (vgpr_total + 7) // 8 * 8)
    return max(1, min(8, 512 // tot)), tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--so", default=str(ROOT / "comet_amd" / "libcomet_hip.so"))
    ap.add_argument("--all", action="store_true", help="every kernel (default: the search-path kernels DESIGN.md names)")
    a = ap.parse_args()
    rows = []
    for co in code_objects(Path(a.so)):
        rows += kernels_of(co)
    names = demangle([k.get("name", "") for k in rows])
    want = re.compile(r"flat_scan_q[rn]_kernel|fast_post_kernel|adc_scan2?_kernel|adc_order|hnsw_(search|insert)_kernel|dist_exact_kernel|coarse_(dot_mfma|pick)|ivf_scan_f16|bm25_dense|rrf_fuse|merge_topk|sel_composites|pq_bound3|pq_encode|km_")
    print(f"# {Path(a.so).name}: {len(rows)} gfx950 kernels; columns from the code objects' amdhsa.kernels metadata (llvm-readelf --notes)")
    print(f"# {'kernel':<78} {'wg':>5} {'v+a':>5} {'agpr':>5} {'sgpr':>5} {'lds B':>7} {'scratch B':>9} {'spill v/s':>9} {'alloc':>6} {'waves/SIMD by regs':>18}")
    seen = set()
    for k, nm in sorted(zip(rows, names), key=lambda t: t[1]):
        short = re.sub(r"\(.*$", "", nm).replace("void ", "").replace("comet::", "")
        m_ = re.match(r"_ZN5comet\d+([a-z0-9_]+?)I(.*?)EEv", short)          # names c++filt does not know (_Float16 arguments): kernel<template digits>
        if m_:
            short = m_.group(1) + "<" + m_.group(2) + ">"
        if short in seen or (not a.all and not want.search(short)):
            continue
        seen.add(short)
        v, ag, sg = int(k.get("vgpr_count", 0)), int(k.get("agpr_count", 0)), int(k.get("sgpr_count", 0))
        w, tot = waves_per_simd(v)
        print(f"  {short[:78]:<78} {k.get('max_flat_workgroup_size', '?'):>5} {v:>5} {ag:>5} {sg:>5} {k.get('group_segment_fixed_size', '?'):>7} {k.get('private_segment_fixed_size', '?'):>9} "
              f"{str(k.get('vgpr_spill_count', 0)) + '/' + str(k.get('sgpr_spill_count', 0)):>9} {tot:>6} {w:>18}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
