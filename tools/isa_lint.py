#!/usr/bin/env python3
"""Off-GPU checks of the code the compiler generated for the demod kernels (r6).

The kernels' correctness rests on two things nothing but the GPU suite used to see:
  * build.sh compiles with the LLVM-internal `-mllvm -structurizecfg-skip-uniform-regions`;
  * two inline-asm blocks carry hand-placed wait states for a gfx940 / gfx950 hazard the compiler's hazard recogniser
    cannot see inside inline asm (a VALU read of an SGPR written by VALU needs two wait states): `spread_words`
    (afsk_demod_phasec.h: v_writelane_b32 fed by ballot SGPRs, opened by `s_nop 1`) and the 12000-baud
    `v_cmp_lt_u16_sdwa` block (afsk_demod_rounds_multi.h, closed by `s_nop 1`).
This module reads the device assembly (`AFSK_KEEP_ASM=<dir> build.sh` leaves one .s per demod translation unit) and
checks what can be checked without a GPU:
  1. every inline-asm block with v_writelane_b32 OPENS with s_nop >= 1, every block with v_cmp_*_sdwa CLOSES with one;
     outside inline asm, no instruction that reads VCC / an SGPR pair as a VALU source directly follows ... (left to the
     compiler's hazard recogniser -- not checked);
  2. per kernel: 0 scratch, 0 VGPR spills, occupancy 2 waves / SIMD, LDS 73,792 B per block;
  3. no ds_bpermute / ds_permute / MFMA / scratch_ / flat_load / global_load instruction anywhere (samples reach the
     wave through LDS-DMA only);
  4. the LDS-DMA ring is there: >= 16 `buffer_load_dwordx4 ... lds` per kernel;
  5. instruction-mix SNAPSHOT (tests/golden/isa_snapshot.json: LDS-DMA, v_sad_u16, v_dot2*, v_alignbyte, DPP counts
     and the line count per translation unit, keyed by the kernel source hash and the compiler): a change of the
     generated code that nobody asked for -- a toolchain bump, a flag -- shows up as a diff here, on the CPU.

    python tools/isa_lint.py <asm dir>              lint, print the report, exit 1 on a violation
    python tools/isa_lint.py <asm dir> --snapshot   (re)write tests/golden/isa_snapshot.json from this build
"""
from __future__ import annotations

import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SNAPSHOT = os.path.join(ROOT, "tests", "golden", "isa_snapshot.json")
LDS_PER_BLOCK = 73792          # 4 waves x kFastWaveLdsProduct (afsk_demod_ring.h)
FORBIDDEN = ("ds_bpermute", "ds_permute", "v_mfma", "v_smfma", "scratch_load", "scratch_store", "flat_load",
             "global_load")
COUNTED = {"lds_dma_x4": r"buffer_load_dwordx4 .*\blds\b", "lds_dma_x1": r"buffer_load_dword .*\blds\b",
           "v_sad_u16": r"\bv_sad_u16\b", "v_dot2": r"\bv_dot2", "v_alignbyte": r"\bv_alignbyte_b32\b",
           "v_pk_sub": r"\bv_pk_sub_", "dpp": r"\b(row_shr|row_bcast|quad_perm|row_mirror|row_half_mirror):?",
           "sdwa": r"_sdwa\b", "v_writelane": r"\bv_writelane_b32\b", "s_waitcnt_vmcnt": r"s_waitcnt vmcnt",
           "global_store": r"\bglobal_store_"}


def toolchain() -> str:
    out = subprocess.run(["hipcc", "--version"], capture_output=True, text=True).stdout
    return "\n".join(l.rstrip() for l in out.splitlines() if l.startswith(("HIP version", "AMD clang version")))


def source_hash() -> str:
    sys.path.insert(0, ROOT)
    import bench
    return bench.kernel_source_hash()


def _instructions(lines):
    """(line number, text) of the real instructions: no labels, directives, comments."""
    for i, l in enumerate(lines, 1):
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        yield i, t.split(";")[0].strip()


def lint_file(path: str) -> dict:
    lines = open(path, errors="replace").read().splitlines()
    name = os.path.basename(path)
    problems = []
    # ---- 1. inline-asm blocks
    blocks, cur = [], None
    for i, l in enumerate(lines, 1):
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            cur = []
        elif t.startswith(";;#ASMEND"):
            if cur is not None:
                blocks.append(cur)
            cur = None
        elif cur is not None and t and not t.startswith(";"):
            cur.append((i, t))
    n_wl = n_sdwa = 0
    for b in blocks:
        text = [t for _, t in b]
        if any(t.startswith("v_writelane_b32") for t in text):
            n_wl += 1
            m = re.match(r"s_nop (\d+)", text[0])
            if not m or int(m.group(1)) < 1:
                problems.append(f"{name}:{b[0][0]}: inline-asm block with v_writelane_b32 does not open with s_nop >= 1 (VALU-written SGPR read by VALU: two wait states on gfx950)")
            if not all(t.startswith(("v_writelane_b32", "s_nop")) for t in text):
                problems.append(f"{name}:{b[0][0]}: unexpected instruction inside the spread_words block")
        if any("_sdwa" in t and t.startswith("v_cmp") for t in text):
            n_sdwa += 1
            m = re.match(r"s_nop (\d+)", text[-1])
            if not m or int(m.group(1)) < 1:
                problems.append(f"{name}:{b[-1][0]}: inline-asm block with v_cmp_*_sdwa does not close with s_nop >= 1")
    # a v_writelane_b32 outside inline asm is the compiler's own (its hazard recogniser pads it): count only
    # ---- 2. per-kernel resources
    kernels = {}
    kname = None
    for l in lines:
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", l)
        if m:
            kname = m.group(1)
            kernels[kname] = {}
        m = re.match(r";\s*(ScratchSize|Occupancy|NumVgprs|NumSgprs|LDSByteSize|codeLenInByte):\s*(\d+)", l)
        if m and kernels:
            # these comment lines follow the kernel body; attach to the most recent kernel without the key
            for k in kernels:
                if m.group(1) not in kernels[k]:
                    kernels[k][m.group(1)] = int(m.group(2))
                    break
    spills = [int(x) for x in re.findall(r"\.vgpr_spill_count:\s*(\d+)", "\n".join(lines))]
    priv = [int(x) for x in re.findall(r"\.private_segment_fixed_size:\s*(\d+)", "\n".join(lines))]
    for k, r in kernels.items():
        if r.get("ScratchSize", 0) != 0:
            problems.append(f"{name}: {k}: scratch {r['ScratchSize']} B")
        if r.get("Occupancy") != 2:
            problems.append(f"{name}: {k}: occupancy {r.get('Occupancy')} waves / SIMD, expected 2 (LDS-limited: 2 blocks of 4 waves per CU)")
        if r.get("LDSByteSize") != LDS_PER_BLOCK:
            problems.append(f"{name}: {k}: LDS {r.get('LDSByteSize')} B per block, expected {LDS_PER_BLOCK}")
    if any(spills):
        problems.append(f"{name}: VGPR spills {spills}")
    if any(priv):
        problems.append(f"{name}: private segment {priv}")
    if not kernels:
        problems.append(f"{name}: no kernel found")
    # ---- 3. / 4. / 5. instruction mix
    counts = {k: 0 for k in COUNTED}
    n_instr = 0
    for i, t in _instructions(lines):
        n_instr += 1
        for f in FORBIDDEN:
            if t.startswith(f):
                problems.append(f"{name}:{i}: forbidden instruction: {t}")
        if t.startswith("buffer_load") and not re.search(r"\blds\b", t):
            problems.append(f"{name}:{i}: a buffer load into registers (samples reach a wave through LDS-DMA only): {t}")
        for k, pat in COUNTED.items():
            if re.search(pat, t):
                counts[k] += 1
    if counts["lds_dma_x4"] < 16 * max(1, len(kernels)):
        problems.append(f"{name}: only {counts['lds_dma_x4']} LDS-DMA ring requests for {len(kernels)} kernel(s)")
    counts["instructions"] = n_instr
    counts["kernels"] = len(kernels)
    counts["asm_blocks_writelane"] = n_wl
    counts["asm_blocks_sdwa"] = n_sdwa
    return {"file": name, "problems": problems, "counts": counts,
            "resources": {k: {x: r.get(x) for x in ("NumVgprs", "NumSgprs", "ScratchSize", "Occupancy", "LDSByteSize")}
                          for k, r in kernels.items()}}


def lint_dir(d: str) -> dict:
    files = sorted(f for f in os.listdir(d) if f.endswith(".s"))
    rep = {"files": {}, "problems": []}
    for f in files:
        r = lint_file(os.path.join(d, f))
        rep["files"][f[:-2]] = r
        rep["problems"] += r["problems"]
    # the hand-written asm blocks must be where the sources put them: spread_words in every multi-slice round
    # (bit_frames 4 ... 64 and 20), the SDWA compare at bit_frames 4
    def blocks(tu, key):
        return rep["files"].get(tu, {}).get("counts", {}).get(key, 0)
    for tu, key in (("afsk_demod_uniform_20", "asm_blocks_writelane"), ("afsk_demod_uniform_4", "asm_blocks_writelane"),
                    ("afsk_demod_uniform_4", "asm_blocks_sdwa"), ("afsk_demod_big", "asm_blocks_writelane"),
                    ("afsk_demod_big", "asm_blocks_sdwa"), ("afsk_demod_small", "asm_blocks_writelane")):
        if tu in rep["files"] and blocks(tu, key) == 0:
            rep["problems"].append(f"{tu}: expected inline-asm block missing ({key}): was the hazard-padded asm replaced?")
    return rep


def snapshot_of(rep: dict) -> dict:
    return {"kernel_source_hash": source_hash(), "toolchain": toolchain(),
            "counts": {tu: r["counts"] for tu, r in rep["files"].items()},
            "resources": {tu: r["resources"] for tu, r in rep["files"].items()}}


def main() -> int:
    if len(sys.argv) < 2:
        print(__doc__)
        return 2
    rep = lint_dir(sys.argv[1])
    for p in rep["problems"]:
        print("PROBLEM:", p)
    tot = {}
    for r in rep["files"].values():
        for k, v in r["counts"].items():
            tot[k] = tot.get(k, 0) + v
    print(f"{len(rep['files'])} translation units, totals: {json.dumps(tot)}")
    if "--snapshot" in sys.argv:
        if rep["problems"]:
            print("not writing a snapshot of a build with problems")
            return 1
        json.dump(snapshot_of(rep), open(SNAPSHOT, "w"), indent=0, sort_keys=True)
        print("wrote", SNAPSHOT)
    return 1 if rep["problems"] else 0


if __name__ == "__main__":
    raise SystemExit(main())
