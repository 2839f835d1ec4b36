"""Static audit of the hand-counted inline-asm pipelines in the NIF MLP kernel (csrc/nif_kernels.hpp).

hipcc treats an asm load's destination as written when the asm statement ends, so under register pressure it may
copy, spill or reuse that register before the data has landed (cdna_hip_programming.md §5.7 item 1). The kernel
names every destination in its wait statements, which pins ORDER but not ALLOCATION, so this test compiles the
device code to assembly (no GPU needed) and checks every kept instantiation of nif_mlp_kernel:
  * between an asm global_load / ds_read and the hand-written s_waitcnt that retires it, no other instruction reads
    or writes the destination registers (straight-line scan per kernel, in-order completion per counter);
  * no compiler-issued vector-memory STORE or ATOMIC is in flight together with the asm loads: stores return out of
    order with respect to loads, so "N operations left" would no longer identify which asm loads have landed.
    Compiler-issued LOADS may overlap them: loads return in order, so a younger load only makes a hand-counted
    "all but the newest N" wait stricter, and the compiler's own waits for its loads (which it counts without
    knowing about the asm loads) are stricter too when older or younger asm loads are outstanding;
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
        if op == "s_waitcnt" and not in_asm:
            # a compiler-issued wait: only a full drain tells anything about the asm loads (it does not know them)
            if re.search(r"vmcnt\(0\)", t):
                vm = []
            if re.search(r"lgkmcnt\(0\)", t):
                lg = []
            continue
        if op == "s_branch":
            # what follows an unconditional branch is not its fall-through: its predecessors are scanned where they
            # lie (the loop-carried sets reach a loop header from the prologue just above it as well)
            vm, lg = [], []
            continue
        if op.startswith("s_"):
            continue
        if vm and re.match(r"(global|scratch|buffer|flat)_(store|atomic)", op) or (vm and op.startswith("scratch_")):
            # a compiler-issued store / atomic (or any scratch access: a spill) joins the queue the hand-written
            # vmcnt(N) counts, and stores are not ordered with loads: "N left" no longer means "the oldest set has landed"
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


def audit_flow(lines):
    """Flow-sensitive companion of `audit`: over the kernel's control-flow graph, which asm-load destination registers
    MAY still be in flight at every instruction (union over the paths that reach it), and does a compiler-generated
    instruction touch one of them? This is what catches a register copy the compiler places at a loop header or a join
    (the linear scan loses its state at branches). A register stops being in flight at a full drain
    (`s_waitcnt vmcnt(0)` / `lgkmcnt(0)`, whoever issued it) or where the source declares it landed: every
    hand-counted wait is followed by `; landed <reg>` comments naming the registers it retires (csrc/nif_kernels.hpp),
    whose COUNTS the linear scan checks."""
    insts = []          # (text, in_asm)
    label_at = {}
    in_asm = False
    for raw in lines:
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True; continue
        if t.startswith(";;#ASMEND"):
            in_asm = False; continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            label_at[m.group(1)] = len(insts); continue
        if in_asm and t.startswith("; landed"):
            insts.append((t, True)); continue
        if not t or t[0] in ";.":
            continue
        insts.append((t, in_asm))
    n = len(insts)
    succ = [[] for _ in range(n)]
    for i, (t, _) in enumerate(insts):
        op = t.split()[0]
        if op == "s_endpgm":
            continue
        if op == "s_branch":
            succ[i] = [label_at[t.split()[1]]]
        elif op.startswith("s_cbranch"):
            succ[i] = [label_at[t.split()[1]]] + ([i + 1] if i + 1 < n else [])
        elif i + 1 < n:
            succ[i] = [i + 1]
    state_in = [None] * n
    state_in[0] = frozenset()
    work = [0]
    bad = {}
    while work:
        i = work.pop()
        st = set(state_in[i])
        t, asm = insts[i]
        op = t.split()[0]
        args = t[len(op):].split(",")
        if asm and (op.startswith("global_load") or op.startswith("ds_read")):
            st |= {("vm" if op.startswith("global") else "lg", r) for r in _regs(args[0])}
        elif t.startswith("; landed"):
            gone = _regs(t[len("; landed"):])
            st = {(k, r) for k, r in st if r not in gone}
        elif op == "s_waitcnt":
            if re.search(r"vmcnt\(0\)", t):
                st = {(k, r) for k, r in st if k != "vm"}
            if re.search(r"lgkmcnt\(0\)", t):
                st = {(k, r) for k, r in st if k != "lg"}
        elif not asm and not op.startswith("s_"):
            used = set()
            for a in args:
                used |= _regs(a)
            hit = used & {r for _, r in st}
            if hit:
                bad[i] = f"'{t[:70]}' touches v{sorted(hit)[0]} while an asm load into it may be in flight"
        out = frozenset(st)
        for j in succ[i]:
            merged = out if state_in[j] is None else (state_in[j] | out)
            if merged != state_in[j]:
                state_in[j] = merged
                work.append(j)
    return [bad[k] for k in sorted(bad)]


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
        flow = audit_flow(body)
        assert not flow, (k, flow[:5])
        m = re.search(re.escape(k) + r":.*?; ScratchSize: (\d+)", meta, re.S)
        # a few dwords of lane constants parked in scratch between layer runs are tolerated: audit() above has already
        # rejected any scratch access while an asm load is in flight (it would count on vmcnt), i.e. inside a k-loop
        assert m and int(m.group(1)) <= 64, (k, "uses scratch", m and m.group(1))


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
    assert audit(extra_load)[0] == []          # a younger LOAD is harmless (in-order return)
    extra_store = [l.replace("v_add_u32_e32 v3, 1, v3", "global_store_dword v[6:7], v50, off") for l in ok]
    assert len(audit(extra_store)[0]) == 1
    early_use = [l.replace("s_waitcnt vmcnt(0)", "s_waitcnt vmcnt(1)") for l in ok]
    assert len(audit(early_use)[0]) == 1
    # the flow-sensitive pass: a copy of an in-flight register at a join that the linear scan cannot see
    flow = """
	;;#ASMSTART
	global_load_dwordx4 v[10:13], v2, s[4:5]
	;;#ASMEND
	s_cbranch_scc1 .LBB0_2
	s_branch .LBB0_3
.LBB0_2:
	v_add_u32_e32 v3, 1, v3
.LBB0_3:
	v_mov_b32_e32 v40, v11
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	;;#ASMSTART
	; landed v[10:13]
	;;#ASMEND
	v_mov_b32_e32 v41, v12
	s_endpgm
""".split("\n")
    assert audit(flow)[0] == [] and len(audit_flow(flow)) == 1 and "v11" in audit_flow(flow)[0]
