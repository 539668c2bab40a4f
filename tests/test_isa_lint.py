"""CPU suite: what can be checked of the GENERATED device code without a GPU (r6; tools/isa_lint.py has the story).

The demod kernels are built with an LLVM-internal code-generation flag and carry two inline-asm blocks whose wait states
the compiler's hazard recogniser cannot see.  Until r6 only the GPU parity suite could notice a toolchain bump breaking
either.  Here the device assembly of every demod translation unit (left by build.sh with AFSK_KEEP_ASM; rebuilt by this
module when it does not belong to the current sources) is linted, compared with a committed instruction-mix snapshot, and
build.sh's escape hatches (AFSK_SAFE_CODEGEN, the validated-toolchain fallback, the gfx950-only guard) are exercised."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402

CSRC = os.path.join(ROOT, "afskmodem_amd", "csrc")
BUILD = os.path.join(CSRC, "build.sh")


def _flags(**env):
    e = dict(os.environ, AFSK_PRINT_FLAGS="1")
    for k in ("AFSK_SAFE_CODEGEN", "AFSK_FAST_CODEGEN", "AFSK_VALIDATED_TOOLCHAIN", "AFSK_ARCH"):
        e.pop(k, None)
    e.update(env)
    return subprocess.run(["bash", BUILD], env=e, capture_output=True, text=True)


@pytest.fixture(scope="module")
def asm_dir(tmp_path_factory):
    """The assembly of the CURRENT sources with the shipped (fast) code generation: what build() left, or a fresh build."""
    d = os.path.join(CSRC, "_asm")
    info = os.path.join(d, "BUILD_INFO")
    want = isa_lint.source_hash()
    if os.path.exists(info) and open(info).read().split() == [want, "fast"]:
        return d
    d = str(tmp_path_factory.mktemp("asm"))
    e = dict(os.environ, AFSK_KEEP_ASM=d, AFSK_OUT=os.path.join(d, "libafsk_lint.so"))
    e.pop("AFSK_SAFE_CODEGEN", None)
    subprocess.check_call(["bash", BUILD], env=e, stdout=subprocess.DEVNULL)
    assert open(os.path.join(d, "BUILD_INFO")).read().split() == [want, "fast"]
    return d


def test_generated_code_passes_the_lint(asm_dir):
    rep = isa_lint.lint_dir(asm_dir)
    assert len(rep["files"]) == 39, sorted(rep["files"])          # 36 uniform rates + run-time geometry + small + big
    assert not rep["problems"], "\n".join(rep["problems"][:20])
    # the hand-written blocks are where the sources put them, hazard padding included (checked block by block above)
    c = {tu: r["counts"] for tu, r in rep["files"].items()}
    assert c["afsk_demod_uniform_4"]["asm_blocks_sdwa"] >= 1 and c["afsk_demod_uniform_4"]["asm_blocks_writelane"] >= 1
    assert c["afsk_demod_uniform_20"]["asm_blocks_writelane"] >= 1
    assert c["afsk_demod_uniform_40"]["asm_blocks_writelane"] == 0   # one symbol per lane: no multi-slice round
    for tu, r in rep["files"].items():
        for k, res in r["resources"].items():
            assert res["ScratchSize"] == 0 and res["Occupancy"] == 2 and res["LDSByteSize"] == isa_lint.LDS_PER_BLOCK, (tu, k, res)


def test_lint_notices_a_dropped_wait_state_and_foreign_instructions(asm_dir, tmp_path):
    """The lint is only worth something if it fails on the breakage it exists for."""
    src = open(os.path.join(asm_dir, "afsk_demod_uniform_4.s")).read()
    lines = src.splitlines()
    # (a) the s_nop that opens a spread_words block is gone
    i = next(k for k, l in enumerate(lines) if l.strip().startswith("v_writelane_b32") and lines[k - 1].strip().startswith("s_nop"))
    doctored = lines[: i - 1] + lines[i:]
    (tmp_path / "a").mkdir()
    (tmp_path / "a" / "afsk_demod_uniform_4.s").write_text("\n".join(doctored))
    rep = isa_lint.lint_dir(str(tmp_path / "a"))
    assert any("does not open with s_nop" in p for p in rep["problems"]), rep["problems"][:5]
    # (b) the s_nop that closes the SDWA compare block is gone
    j = next(k for k, l in enumerate(lines) if "v_cmp_lt_u16_sdwa" in l)
    while not lines[j].strip().startswith("s_nop"):
        j += 1
    (tmp_path / "b").mkdir()
    (tmp_path / "b" / "afsk_demod_uniform_4.s").write_text("\n".join(lines[:j] + lines[j + 1:]))
    rep = isa_lint.lint_dir(str(tmp_path / "b"))
    assert any("does not close with s_nop" in p for p in rep["problems"]), rep["problems"][:5]
    # (c) a cross-lane shuffle through the LDS pipe, scratch traffic, a sample load into registers
    k = next(k for k, l in enumerate(lines) if l.strip().startswith("ds_read_b128"))
    for bad, what in (("\tds_bpermute_b32 v1, v2, v3", "forbidden"), ("\tscratch_store_dword off, v1, s0", "forbidden"),
                      ("\tbuffer_load_dwordx4 v[0:3], v4, s[8:11], 0 offen", "buffer load into registers")):
        (tmp_path / "c").mkdir(exist_ok=True)
        (tmp_path / "c" / "afsk_demod_uniform_4.s").write_text("\n".join(lines[:k] + [bad] + lines[k:]))
        rep = isa_lint.lint_dir(str(tmp_path / "c"))
        assert any(what in p for p in rep["problems"]), (bad, rep["problems"][:5])
    # (d) scratch / occupancy in the kernel's resource comments
    (tmp_path / "d").mkdir()
    (tmp_path / "d" / "afsk_demod_uniform_4.s").write_text(src.replace("; ScratchSize: 0", "; ScratchSize: 64", 1).replace("; Occupancy: 2", "; Occupancy: 1", 1))
    rep = isa_lint.lint_dir(str(tmp_path / "d"))
    assert any("scratch 64" in p for p in rep["problems"]) and any("occupancy 1" in p for p in rep["problems"])


def test_instruction_mix_equals_the_committed_snapshot(asm_dir):
    """tests/golden/isa_snapshot.json is written (tools/isa_lint.py <dir> --snapshot) from the build the GPU suite last
    validated.  Same sources + same compiler must give the same code; sources that changed since must be validated on
    the GPU and the snapshot refreshed BEFORE the commit -- this test is the reminder."""
    snap = json.load(open(isa_lint.SNAPSHOT))
    if snap["toolchain"] != isa_lint.toolchain():
        pytest.skip("another compiler than the snapshot's: nothing to compare (build.sh falls back to the default code generation)")
    assert snap["kernel_source_hash"] == isa_lint.source_hash(), (
        "the demod kernel sources (or build.sh) changed since tests/golden/isa_snapshot.json was written: run the GPU "
        "suite on this build, then `python tools/isa_lint.py afskmodem_amd/csrc/_asm --snapshot`")
    rep = isa_lint.lint_dir(asm_dir)
    now = {tu: r["counts"] for tu, r in rep["files"].items()}
    diff = {tu: {k: (snap["counts"][tu].get(k), v) for k, v in c.items() if snap["counts"].get(tu, {}).get(k) != v}
            for tu, c in now.items() if snap["counts"].get(tu) != c}
    assert not diff, f"generated code differs from the snapshot (snapshot, now): {json.dumps(diff)[:1500]}"


def test_build_switches():
    """AFSK_SAFE_CODEGEN drops the LLVM-internal flag; so does a compiler other than the validated one (unless
    AFSK_FAST_CODEGEN insists); any target but gfx950 is refused."""
    flag = "-structurizecfg-skip-uniform-regions"
    r = _flags()
    assert r.returncode == 0 and r.stdout.startswith("fast:") and flag in r.stdout, r.stdout + r.stderr
    r = _flags(AFSK_SAFE_CODEGEN="1")
    assert r.returncode == 0 and r.stdout.startswith("safe:") and flag not in r.stdout and "-mllvm" not in r.stdout
    other = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"afsk_vt_{os.getpid()}")
    open(other, "w").write("# another compiler\nHIP version: 0.0\nAMD clang version 0.0\n")
    try:
        r = _flags(AFSK_VALIDATED_TOOLCHAIN=other)
        assert r.returncode == 0 and r.stdout.startswith("safe:") and flag not in r.stdout and "WARNING" in r.stderr
        r = _flags(AFSK_VALIDATED_TOOLCHAIN=other, AFSK_FAST_CODEGEN="1")
        assert r.returncode == 0 and r.stdout.startswith("fast:") and flag in r.stdout
    finally:
        os.unlink(other)
    for arch in ("gfx942", "gfx90a", "gfx1100"):
        r = _flags(AFSK_ARCH=arch)
        assert r.returncode != 0 and "refused" in r.stderr
    # the file build.sh compares with names the compiler of THIS image (the one the r6 GPU visits ran)
    want = [l.strip() for l in open(os.path.join(CSRC, "VALIDATED_TOOLCHAIN")) if l.strip() and not l.startswith("#")]
    assert want == isa_lint.toolchain().splitlines()
    assert shutil.which("hipcc")
