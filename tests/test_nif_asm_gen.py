"""Replay of K3a's (and K3b's: the four-wave shape) generated instruction stream (csrc/nif_asm_gen.py; no GPU): every statement of the
hand-scheduled dense stack is checked against what the dataflow needs, the way tests/test_asm_pipeline_audit.py checks K3's hand-counted
waits.

The replay keeps, per register, WHAT it holds (which weight fragment, which bias tile, which activation) and whether the LDS read
that fills it has been retired by a counted wait, and walks the text:
  * every v_mfma takes as A the fragment the packed stream (NifRegsDevice::load) has next for it - landed, read from the chunk slot
    the wave was told to read -, as B the k-step of its layer's input that belongs to it (an activation register written by the
    epilogue of the right pair of the previous layer, ReLU applied where the layer has one; or the feature operand), as C the
    landed bias of its tile in its first k-step and its own accumulator afterwards;
  * no instruction writes a register an outstanding LDS read will fill, and none reads one before the wait that retires it;
  * an epilogue convert reads an accumulator only once all its MFMAs were issued at least four MFMAs earlier (the wait states an
    MFMA result needs before a VALU read are far shorter), and before the next pair that uses the buffer writes it;
  * the first read of a chunk comes behind the wave's s_waitcnt vmcnt(0) + s_barrier for it, every LDS-DMA piece of a chunk is issued
    behind the entry of the chunk before it (when the slot it lands in - the slot of the chunk two back - is no longer read by
    anybody) and in front of the chunk's own entry, every wave issues its share of every chunk exactly once;
  * lgkmcnt never has to hold more than 15."""
import re
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ipu_ray_lib_amd" / "csrc"))
import nif_asm_gen as gen      # noqa: E402


def regs_of(tok):
    """'v[8:11]' / 'a[0:3]' / 'v7' -> [('v', 8), ...]; None for anything else (operands, constants)."""
    m = re.match(r"([va])\[(\d+):(\d+)\]$", tok)
    if m:
        return [(m.group(1), k) for k in range(int(m.group(2)), int(m.group(3)) + 1)]
    m = re.match(r"([va])(\d+)$", tok)
    return [(m.group(1), int(m.group(2)))] if m else None


def replay(kinds, relu, mt=2, nw=8, **kw):
    gen.NW = nw
    try:
        net, lines = gen.generate(kinds, relu, mt=mt, **kw)
    finally:
        gen.NW = 8
    L = gen.Lay(mt, nw)
    HT, D = gen.HT, gen.D
    pieces_of = lambda cnt: len([p for p in range(40) if nw * p < cnt])      # noqa: E731
    # what the stream holds: consumed fragments in order
    expect = []
    for c, (first, cnt) in enumerate(net.chunks):
        for q in range(cnt):
            f = net.frags[first + q]
            if f is not None:
                expect.append((c, q) + f)
    holds = {}                 # register -> content
    pending = []               # outstanding LDS reads: (dest registers, content)
    cur_chunk_read = None      # chunk whose slot the address register points at
    entered = set()            # chunks whose entry (vmcnt(0) + barrier) this wave has passed
    dma = {}                   # chunk (absolute, may be nchunks = next pass's 0) -> pieces issued
    n_mfma = 0
    acc_state = {}             # accumulator base register -> dict(l, j, tt, m, done k-steps, last mfma index)
    out_bias = {}              # K3b, final layer: output operand -> bias offset loaded into it
    act = {}                   # activation register -> (layer, j, m, idx, relu applied)
    tmp = {}                   # K3b: temporary -> (layer, j, m, idx, relu applied) on its way to the accumulator file
    step_i = 0                 # index into expect (each consumed mt times)
    m_next = 0
    drained = barrier_after_drain = False
    last_m0 = None
    nchunks = len(net.chunks)
    VA = ("v", L.vaddr())
    acc_lo, acc_hi = gen.ACC, gen.ACC + 2 * L.acc_stride

    def in_flight(regs):
        return [p for p in pending if set(p[0]) & set(regs)]

    def act_base(which, ks, m):
        f, base = L.act(which, ks, m)
        return [(f, base + k) for k in range(4)]

    for ln, t in enumerate(lines):
        if t.startswith(";"):
            continue
        op, _, rest = t.partition(" ")
        args = [a.strip() for a in rest.split(",")] if rest else []
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                assert n <= 15
                while len(pending) > n:
                    dest, content = pending.pop(0)
                    for r in dest:
                        holds[r] = content
            if "vmcnt(0)" in t:
                drained = True; barrier_after_drain = False
            continue
        if op == "s_barrier":
            assert drained, f"line {ln}: a barrier without the wave's pieces waited for"
            barrier_after_drain = True
            continue
        if op == "v_add_u32" and regs_of(args[0]) == [VA]:
            # the wave turns to the next chunk's slot: behind vmcnt(0) + barrier
            assert barrier_after_drain, f"line {ln}: chunk entry without wait + barrier"
            cur_chunk_read = 0 if cur_chunk_read is None else cur_chunk_read + 1
            entered.add(cur_chunk_read)
            # every piece of this chunk must have been issued (by this wave) before its entry - except the first chunk of the pass,
            # which the previous pass (or the kernel's prologue) fetched
            if cur_chunk_read > 0:
                want = pieces_of(net.chunks[cur_chunk_read][1])
                assert dma.get(cur_chunk_read, 0) == want, (cur_chunk_read, dma.get(cur_chunk_read), want)
            drained = False
            continue
        if op == "ds_read_b128":
            off = int(re.search(r"offset:(\d+)", t).group(1))
            addr = args[1].split()[0]
            if regs_of(addr) == [VA]:
                dest = regs_of(args[0])
                assert not in_flight(dest), f"line {ln}: read into registers with a read in flight"
                assert cur_chunk_read in entered
                content = ("frag", cur_chunk_read, off // 1024)
                assert off // 1024 < net.chunks[cur_chunk_read][1]
            else:
                assert addr == "%[biasv]"
                content = ("bias", off)
                dest = regs_of(args[0])
                assert dest == [(L.bias(0)[0], L.bias(0)[1] + k) for k in range(4)] or dest == [(L.bias(1)[0], L.bias(1)[1] + k) for k in range(4)]
                assert not in_flight(dest), f"line {ln}: read into registers with a read in flight"
            pending.append((dest, content))
            assert len(pending) <= 15
            continue
        if op == "s_add_u32" and args[0] == "m0":
            last_m0 = int(args[2])
            continue
        if op == "global_load_lds_dwordx4":
            # which chunk: from the source offset of the s_add in front of it
            src = int(re.search(r", (\d+)$", lines[ln - 4]).group(1)) // 1024
            tgt = [c for c, (first, cnt) in enumerate(net.chunks) if first <= src < first + cnt][0]
            p = (src - net.chunks[tgt][0]) // nw
            assert (src - net.chunks[tgt][0]) % nw == 0 and last_m0 == nw * 1024 * p
            absolute = tgt if tgt > (cur_chunk_read or 0) else tgt + nchunks
            assert absolute == cur_chunk_read + 1, f"line {ln}: a piece of chunk {absolute} issued while chunk {cur_chunk_read} is being read"
            assert dma.get(absolute, 0) == p, "pieces in order, each once"
            dma[absolute] = p + 1
            continue
        if op == "v_mfma_f32_16x16x32_f16":
            d_, a_, b_, c_ = args
            c, q, l, j, ks, tt = expect[step_i]
            m = m_next
            kind = net.kinds[l]
            last = kind in "LM"
            ra = regs_of(a_)
            assert ra == [(L.ring(step_i % D)[0], L.ring(step_i % D)[1] + k) for k in range(4)]
            assert not in_flight(ra), f"line {ln}: MFMA reads a fragment that has not landed"
            assert all(holds.get(r) == ("frag", c, q) for r in ra), f"line {ln}: A operand holds {holds.get(ra[0])}, wanted fragment {(c, q)}"
            act_steps = 0 if kind == "F" else HT
            if ks < act_steps:
                rb = regs_of(b_)
                assert rb == act_base(0 if l % 2 == 1 else 1, ks, m), (ln, b_)
                for k, r in enumerate(rb):
                    assert act.get(r) == (l - 1, ks, m, k, bool(relu[l - 1])), f"line {ln}: B register {r} holds {act.get(r)}"
            else:
                assert b_ == f"%[f{ks - act_steps}{m}]", (ln, b_)
            tile = 0 if last else 2 * j + tt
            want_bias = ("bias", 4 * net.bias_base[l] + 64 * tile)
            if last and mt == 2:
                assert d_ == f"%[o{m}]"
                rd_ = [("o", m)]
            elif last:
                rd_ = regs_of(d_)
                assert rd_ == [("v", gen.ACC + 4 * m + k) for k in range(4)], "K3b's outputs are tied to v[160:175]"
                st0 = acc_state.get(rd_[0])
                assert st0 is None or st0.get("converted") or ks > 0, f"line {ln}: the final layer writes an accumulator that is not converted yet"
            else:
                rd_ = regs_of(d_)
                assert rd_[0][0] == "v" and acc_lo <= rd_[0][1] < acc_hi
            if ks == 0:
                rc = regs_of(c_)
                assert rc[0][0] == "v", "an MFMA's C and D operands share a register file"
                assert not in_flight(rc)
                assert all(holds.get(r) == want_bias for r in rc), f"line {ln}: C operand {holds.get(rc[0])}"
                if not last:
                    st = acc_state.get(rd_[0])
                    assert st is None or st.get("converted"), f"line {ln}: accumulator rewritten before its epilogue"
                    acc_state[rd_[0]] = dict(l=l, j=j, tt=tt, m=m, steps=1, last=n_mfma, converted=False)
            else:
                assert c_ == d_
                if not last:
                    st = acc_state[rd_[0]]
                    assert (st["l"], st["j"], st["tt"], st["m"]) == (l, j, tt, m) and st["steps"] == ks
                    st["steps"] += 1; st["last"] = n_mfma
            n_mfma += 1
            m_next = (m_next + 1) % mt
            if m_next == 0:
                step_i += 1
            continue
        if op == "v_cvt_pk_f16_f32":
            dst = regs_of(args[0])[0]; a0 = regs_of(args[1])[0]; a1 = regs_of(args[2])[0]
            assert a1 == (a0[0], a0[1] + 1) and a0[0] == "v"
            base = ("v", gen.ACC + ((a0[1] - gen.ACC) // 4) * 4)
            st = acc_state[base]
            assert st["steps"] == net.ks[st["l"]], f"line {ln}: convert of an unfinished accumulator"
            assert n_mfma - st["last"] > 4, f"line {ln}: convert {n_mfma - st['last']} MFMAs behind the accumulator's last write"
            half = (a0[1] - base[1]) // 2
            idx = 2 * st["tt"] + half
            final = act_base(0 if st["l"] % 2 == 0 else 1, st["j"], st["m"])[idx]
            if final[0] == "v":
                assert dst == final, (ln, dst, final)
                act[dst] = (st["l"], st["j"], st["m"], idx, False)
            else:
                assert dst[0] == "v" and 224 <= dst[1] < 228, f"line {ln}: a convert for the accumulator-file set goes through a temporary"
                tmp[dst] = (st["l"], st["j"], st["m"], idx, False)
            st.setdefault("cv", set()).add(half)
            if len(st["cv"]) == 2:
                st["converted"] = True
            continue
        if op == "v_pk_max_f16":
            dst = regs_of(args[0])[0]
            assert regs_of(args[1])[0] == dst and args[2] == "0"
            where = act if dst in act and not (224 <= dst[1] < 228 and dst[0] == "v" and mt != 2) else tmp
            l0, j0, m0_, k0, r0 = where[dst]
            assert relu[l0] and not r0
            where[dst] = (l0, j0, m0_, k0, True)
            continue
        if op == "v_accvgpr_write_b32":
            dst = regs_of(args[0])[0]; src = regs_of(args[1])[0]
            l0, j0, m0_, k0, r0 = tmp.pop(src)
            assert r0 == bool(relu[l0]), f"line {ln}: moved before its ReLU"
            assert dst == act_base(1, j0, m0_)[k0] and l0 % 2 == 1
            act[dst] = (l0, j0, m0_, k0, r0)
            continue
        if op in ("s_add_u32", "s_addc_u32", "s_cmp_lt_u32", "s_cselect_b32", "s_mov_b32", "s_sub_u32", "s_nop"):
            continue
        raise AssertionError(f"line {ln}: unexpected instruction {t}")
    assert step_i == len(expect) and not pending and not tmp
    assert n_mfma == mt * len(expect)
    # the pieces of the next pass's chunk 0 went out in this pass's last chunk
    assert dma.get(nchunks) == pieces_of(40)
    return n_mfma, len(lines)


@pytest.mark.parametrize("kinds", ["FPPCPPL", "FPL", "FCPM", "FPPPPPPL"])
def test_generated_body_replays(kinds):
    relu = [True] * (len(kinds) - 1) + [False]
    n_mfma, n_lines = replay(kinds, relu)
    per_tile = {"F": 40, "P": 200, "C": 240, "L": 10, "M": 12}
    assert n_mfma == 2 * sum(per_tile[k] for k in kinds)


@pytest.mark.parametrize("kinds", ["FPPCPPL", "FPL", "FCPM"])
def test_generated_body_of_the_four_wave_shape_replays(kinds):
    """K3b: four waves of 64 rays, one activation set and the fragment ring in the accumulator file."""
    relu = [True] * (len(kinds) - 1) + [False]
    n_mfma, n_lines = replay(kinds, relu, mt=4, nw=4)
    per_tile = {"F": 40, "P": 200, "C": 240, "L": 10, "M": 12}
    assert n_mfma == 4 * sum(per_tile[k] for k in kinds)


def test_generated_body_other_placements_replay():
    relu = [True, True, False, True, True, True, False]
    replay("FPPCPPL", relu, dma_at=(2, 5, 9, 14, 20), epi_start=6)
    replay("FPPCPPL", relu, mt=4, nw=4, epi_start=6)


HIPCC = __import__("shutil").which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not available")
def test_compiled_statements_keep_their_operands_out_of_the_named_registers(tmp_path):
    """What the replay cannot see: where hipcc PUT the statement's operands. The bodies name v0 - v216 (K3a) / v0 - v236 and a0 - a175
    (K3b) literally; every operand the compiler allocates - the feature fragments, the two addresses, K3a's outputs - must lie outside
    them (they are clobbers, or early-clobber tied outputs: an input that shares a register with one is overwritten in the first lines
    of the statement - that happened once, with a memory fault for a result). Compiles the device code to assembly (no GPU) and reads
    the operands back from the first lines of each statement; also: no scratch access and no compiler instruction inside a statement."""
    import subprocess
    out = tmp_path / "raylib.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-Wno-unused-function",
           "-I", str(ROOT / "include"), "-S", "--cuda-device-only", "-o", str(out), str(ROOT / "ipu_ray_lib_amd" / "csrc" / "raylib.hip")]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    text = out.read_text()
    seen = 0
    for mt, last_v, last_a in ((2, gen.Lay(2, 8).last_v, -1), (4, gen.Lay(4, 4).last_v, gen.Lay(4, 4).last_a)):
        m = re.search(r"^(_ZN2mi14nif_asm_kernelILj%dE\w*):" % mt, text, re.M)
        assert m, f"nif_asm_kernel<{mt}> not found"
        body = text[text.index(m.group(1) + ":"):]
        body = body[:body.index(".Lfunc_end")].split("\n")
        starts = [i for i, l in enumerate(body) if "ASMSTART" in l]
        ends = [i for i, l in enumerate(body) if "ASMEND" in l]
        a, b = max(zip(starts, ends), key=lambda t: t[1] - t[0])
        stmt = [l.strip() for l in body[a + 1:b] if l.strip() and not l.strip().startswith(";")]
        assert len(stmt) > 6000
        named_v, named_a = set(range(last_v + 1)), set(range(last_a + 1))

        def outside(tok, what):
            r = regs_of(tok.split()[0])
            assert r, (what, tok)
            for f, k in r:
                assert k not in (named_v if f == "v" else named_a), f"nif_asm_kernel<{mt}>: operand {what} = {tok} lies in the statement's named registers"

        # lane16: source of the first v_add_u32 (chunk address) and of every LDS-DMA; biasv: address of the first bias read
        va = next(l for l in stmt if l.startswith("v_add_u32"))
        outside(va.split(",")[2].strip(), "lane16")
        dma = next(l for l in stmt if l.startswith("global_load_lds_dwordx4"))
        outside(dma.split()[1].rstrip(","), "lane16 (LDS-DMA)")
        vaddr = va.split()[1].rstrip(",")
        bias = next(l for l in stmt if l.startswith("ds_read_b128") and l.split(",")[1].split()[0] != vaddr)
        outside(bias.split(",")[1].split()[0], "biasv")
        # the feature fragments: B operands of the first layer's MFMAs (2 k-steps x mt ray tiles)
        mf = [l for l in stmt if l.startswith("v_mfma")][:4 * mt]
        for l in mf:
            outside(l.split(",")[2].strip(), "a feature fragment")
        if mt == 2:
            for l in [l for l in stmt if l.startswith("v_mfma")][-2:]:
                outside(l.split(",")[0].split(None, 1)[1].strip(), "an output")
        assert not [l for l in stmt if l.startswith("scratch_") or l.startswith("buffer_")], "scratch access inside the statement"
        seen += 1
    assert seen == 2
