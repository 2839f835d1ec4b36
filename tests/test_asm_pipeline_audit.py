"""Static audit of the hand-counted inline-asm pipelines in the NIF MLP kernel (csrc/nif_kernels.hpp).

hipcc treats an asm load's destination as written when the asm statement ends, so under register pressure it may
copy, spill or reuse that register before the data has landed (cdna_hip_programming.md §5.7 item 1). The kernel
names every destination in its wait statements, which pins ORDER but not ALLOCATION, so this test compiles the
device code to assembly (no GPU needed) and checks every kept instantiation of nif_mlp_kernel:
  * between an asm global_load / ds_read and the hand-written s_waitcnt that retires it, no other instruction reads
    or writes the destination registers (straight-line scan per kernel, in-order completion per counter);
  * no compiler-issued vector-memory instruction is in flight together with the asm loads (it would shift the
    hand-written vmcnt counts);
  * the kernels use no scratch (a spill of an in-flight destination would not show as a register access)."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _regs(tok):
    tok = tok.strip().split()[0] if tok.strip() else ""
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def audit(lines):
    """Returns (violations, asm global loads seen, asm ds reads seen)."""
    vm, lg, bad, in_asm, n_vm, n_lg = [], [], [], False, 0, 0
    for i, raw in enumerate(lines):
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True; continue
        if t.startswith(";;#ASMEND"):
            in_asm = False; continue
        if not t or t[0] in ";.":
            continue
        op = t.split()[0]
        args = t[len(op):].split(",")
        if in_asm and op.startswith("global_load"):
            vm.append((_regs(args[0]), i)); n_vm += 1; continue
        if in_asm and op.startswith("ds_read"):
            lg.append((_regs(args[0]), i)); n_lg += 1; continue
        if op == "s_waitcnt" and in_asm:
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1)); vm = vm[len(vm) - n:] if 0 < n < len(vm) else ([] if n == 0 else vm)
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1)); lg = lg[len(lg) - n:] if 0 < n < len(lg) else ([] if n == 0 else lg)
            continue
        if op == "s_branch":
            # what follows an unconditional branch is not its fall-through: its predecessors are scanned where they
            # lie (the loop-carried sets reach a loop header from the prologue just above it as well)
            vm, lg = [], []
            continue
        if op.startswith("s_"):
            continue
        if vm and re.match(r"(global|scratch|buffer|flat)_(load|store|atomic)", op):
            # a compiler-issued VMEM operation joins the in-order queue the hand-written vmcnt(N) counts: with one
            # more operation outstanding, "N left" no longer means "the oldest set has landed"
            bad.append(f"line {i + 1}: compiler-issued '{t[:60]}' while asm loads from line {vm[0][1] + 1} are in flight")
        used = set()
        for a in args:
            used |= _regs(a)
        for pend in (vm, lg):
            for dest, ln in pend:
                if used & dest:
                    bad.append(f"line {i + 1}: '{t[:80]}' touches v{sorted(used & dest)[0]} of the load issued at line {ln + 1}")
                    break
    return bad, n_vm, n_lg


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not available")
def test_nif_kernel_asm_loads_are_not_touched_before_their_wait(tmp_path):
    out = tmp_path / "raylib.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-Wno-unused-function",
           "-I", str(ROOT / "include"), "-S", "--cuda-device-only", "-o", str(out), str(ROOT / "ipu_ray_lib_amd" / "csrc" / "raylib.hip")]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    text = out.read_text().split("\n")
    # split into kernels
    kernels, name, body = {}, None, []
    for l in text:
        m = re.match(r"^(_ZN2mi14nif_mlp_kernel\w+):", l)
        if m:
            name, body = m.group(1), []
        if name:
            body.append(l)
            if "s_endpgm" in l:
                kernels[name] = body; name = None
    assert len(kernels) >= 8, sorted(kernels)
    meta = "\n".join(text)
    for k, body in kernels.items():
        bad, n_vm, n_lg = audit(body)
        assert n_vm > 0 and n_lg > 0, (k, "no asm loads found: the audit no longer matches the kernel")
        assert not bad, (k, bad[:5])
        m = re.search(re.escape(k) + r":.*?; ScratchSize: (\d+)", meta, re.S)
        assert m and int(m.group(1)) == 0, (k, "uses scratch", m and m.group(1))


def test_audit_flags_what_it_should():
    """The scanner itself, on hand-written snippets."""
    ok = """
\t;;#ASMSTART
\tglobal_load_dwordx4 v[10:13], v2, s[4:5]
\t;;#ASMEND
\tv_add_u32_e32 v3, 1, v3
\t;;#ASMSTART
\ts_waitcnt vmcnt(0)
\t;;#ASMEND
\tv_mfma_f32_16x16x32_f16 v[20:23], v[10:13], v[30:33], v[20:23]
""".split("\n")
    assert audit(ok)[0] == []
    touched = [l.replace("v_add_u32_e32 v3, 1, v3", "v_mov_b32_e32 v40, v11") for l in ok]
    assert len(audit(touched)[0]) == 1
    extra_load = [l.replace("v_add_u32_e32 v3, 1, v3", "global_load_dword v50, v[6:7], off") for l in ok]
    assert len(audit(extra_load)[0]) == 1
    early_use = [l.replace("s_waitcnt vmcnt(0)", "s_waitcnt vmcnt(1)") for l in ok]
    assert len(audit(early_use)[0]) == 1
