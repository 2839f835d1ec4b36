#!/usr/bin/env python3
"""nif_asm_gen.py - generator of K3a's hand-scheduled body (csrc/nif_asm_kernel.hpp includes its output).

K3a is K3r's dataflow (csrc/nif_regs_kernel.hpp: a wave owns 32 rays for the whole network and keeps their activations
in registers; the packed weight stream goes through a three-slot LDS ring by LDS-DMA once per 256 rays) with every
instruction of the dense stack placed by hand: one asm statement per workgroup pass holds all the layers. hipcc only
compiles what surrounds it (Fourier features into registers, decode, stores). Reference mathematics:
src/neural_networks/NifModel.cpp:300-327 (dense stack, input re-concatenated where the widths ask for it).

What is placed, per wave (8 waves = 2 per SIMD, 256 registers each):
  * per weight fragment (1 KiB, A operand, read from the ring by ds_read_b128 D fragments ahead): two
    v_mfma_f32_16x16x32_f16 (the wave's two 16-ray tiles); the read of fragment f + D sits right behind the second MFMA
    of fragment f, the counted s_waitcnt lgkmcnt(N) in front of the first MFMA of the fragment it retires (the counts come
    from a replay of the LDS queue in this generator, not from hand);
  * a pair of output tiles ends in 8 v_cvt_pk_f16_f32 + 8 v_pk_max_f16 (ReLU on the rounded halves) that write the next
    layer's B operands; they are dealt one per MFMA into the NEXT pair's MFMAs (accumulators are double-buffered), so no
    epilogue ever stands between two MFMAs of its own wave;
  * the LDS-DMA pieces of the next chunk (5 per wave and 40-KiB chunk) one at a time at fixed fragment positions of the
    first half of a chunk, each 4 scalar instructions + the DMA, never in a burst;
  * ONE s_waitcnt vmcnt(0) + s_barrier per chunk, placed D fragments BEFORE the chunk's first fragment is consumed (where
    its first read is issued), so the fragment ring never drains at a chunk boundary: at that point the wave's pieces of
    the chunk were issued at least half a chunk earlier;
  * nothing else: no address arithmetic in vector registers besides one v_add per chunk, no scratch, no branch.

Register plan (v0 - v216 are named literally and listed as clobbers; operands live above them):
  v[0:79]    activations, set A: k-step ks, ray tile m at 8 ks + 4 m  (4 registers = 8 halves)
  v[80:159]  activations, set B (a layer reads one set and writes the other)
  v[160:191] accumulators, two buffers of [tile of the pair][ray tile][4]
  v[192:207] fragment ring, D = 4
  v[208:215] the pair's bias (C operand of its first MFMAs), [tile][4]
  v216       LDS address of the chunk being read (slot base + 16 lane)
  s[40:47]   ring bookkeeping (clobbered)

Usage: nif_asm_gen.py OUT.inc [--kinds FPPCPPL] [--relu 1111110] - writes the asm text as C string literals plus
`#define`s with the clobber list and the network shape the text was generated for."""
import argparse
import sys

HT = 10                      # hidden width / 32
CH = 4 * HT                  # fragments per chunk
SLOT = CH * 1024
RING_SLOTS = 3
D = 4                        # fragment ring depth (set_ring_depth)
NW = 8                       # waves per workgroup (tools/k3a_lab.py also times a 4-wave form: one wave per SIMD)

SETA, SETB, ACC, RING = 0, 80, 160, 192
BIAS, VADDR, LAST_VGPR = RING + 4 * D, RING + 4 * D + 8, RING + 4 * D + 8
S_RD, S_WR, S_SRC0, S_SRC1, S_T0, S_T1, S_END, S_T2 = 40, 41, 42, 43, 44, 45, 46, 47


def set_ring_depth(d):
    """(experiments: tools/k3a_lab.py) the registers behind the fragment ring move with its depth"""
    global D, BIAS, VADDR, LAST_VGPR
    D = d
    BIAS, VADDR, LAST_VGPR = RING + 4 * D, RING + 4 * D + 8, RING + 4 * D + 8
    assert LAST_VGPR <= 224, "the statement's operands need the registers above"


def vr(base, n=4):
    return f"v[{base}:{base + n - 1}]" if n > 1 else f"v{base}"


class Net:
    def __init__(self, kinds, relu):
        assert kinds[0] == "F" and kinds[-1] in "LM" and all(k in "PC" for k in kinds[1:-1])      # M = last layer with concat
        self.kinds, self.relu = kinds, relu
        self.ks = [2 if k == "F" else HT + 2 if k in "CM" else HT for k in kinds]
        # the stream exactly as NifRegsDevice::load lays it out (nif_regs_pack.hpp): per hidden layer pair j, k-step ks, tile
        # 2j then 2j + 1; the final layer its one tile k-step by k-step, zero fragments up to a multiple of 8; chunks of CH
        self.frags = []          # (layer, j, ks, tt) or None for padding
        self.chunks = []         # (first fragment, count)
        self.bias_base = []      # floats
        nb = 0
        for l, k in enumerate(kinds):
            first = len(self.frags)
            self.bias_base.append(nb)
            if k in "LM":
                nb += 16
                for ks in range(self.ks[l]):
                    self.frags.append((l, 0, ks, 0))
            else:
                nb += 32 * HT
                for j in range(HT):
                    for ks in range(self.ks[l]):
                        self.frags.append((l, j, ks, 0)); self.frags.append((l, j, ks, 1))
            while (len(self.frags) - first) % 8:
                self.frags.append(None)
            n = len(self.frags) - first
            for at in range(0, n, CH):
                self.chunks.append((first + at, min(CH, n - at)))
        self.bias_floats = nb


class Emit:
    def __init__(self):
        self.items = []          # ("i", text) | ("read", tag, text) | ("wait", tag)

    def i(self, text): self.items.append(("i", text))
    def read(self, tag, text): self.items.append(("read", tag, text))
    def wait(self, tag): self.items.append(("wait", tag))

    def render(self):
        """Replays the LDS queue (ds_read complete in issue order) and turns every wait-for-tag into the count that retires it."""
        out, q = [], []
        for it in self.items:
            if it[0] == "i":
                out.append(it[1])
            elif it[0] == "read":
                q.append(it[1]); out.append(it[2])
                assert len(q) <= 15, "lgkmcnt holds 4 bits"
            else:
                if it[1] in q:
                    after = len(q) - q.index(it[1]) - 1
                    out.append(f"s_waitcnt lgkmcnt({after})")
                    q = q[len(q) - after:] if after else []
        if q:
            out.append("s_waitcnt lgkmcnt(0)")      # (only the timing-only builds that drop waits leave anything here)
        return out


class Lay:
    """Where things live, for the two workgroup shapes.
    K3a (MT = 2 ray tiles per wave, 8 waves = two per SIMD, 256 registers): everything in architectural registers (header).
    K3b (MT = 4, 4 waves = one per SIMD, 512 registers = 256 architectural + 256 accumulator-file): activation set A in v[0:159], set
    B in a[0:159] (an MFMA takes its B operand from either file; a layer that writes set B converts into temporaries and moves them
    with v_accvgpr_write), accumulators v[160:223] (two buffers of [tile][ray tile][4]), four temporaries v[224:227], the pair's bias
    v[228:235] (an MFMA's C and D operands share one register FILE, so the bias sits beside the accumulators), the chunk address v236,
    the fragment ring a[160:175] (ds_read_b128 fills accumulator-file registers directly; an MFMA reads its A operand from them).
    Every A fragment read from LDS feeds FOUR MFMAs: half K3a's LDS bytes per MFMA."""

    def __init__(self, mt, nw):
        self.mt, self.nw = mt, nw
        if mt == 2:
            self.acc_stride = 16
            self.last_v, self.last_a = LAST_VGPR, -1
        else:
            self.acc_stride = 32
            self.last_v, self.last_a = 236, 175

    def reg(self, f, base, n=4):
        return f"{f}[{base}:{base + n - 1}]" if n > 1 else f"{f}{base}"

    def act(self, which, ks, m):            # which: 0 = set A, 1 = set B -> (file, first register) of k-step ks, ray tile m
        if self.mt == 2:
            return ("v", (SETA if which == 0 else SETB) + 8 * ks + 4 * m)
        return ("v" if which == 0 else "a", 16 * ks + 4 * m)

    def acc(self, buf, tt, m):
        return ("v", ACC + self.acc_stride * buf + 4 * (self.mt * tt + m))

    def ring(self, i):
        return ("v", RING + 4 * i) if self.mt == 2 else ("a", 160 + 4 * i)

    def bias(self, tt):
        return ("v", BIAS + 4 * tt) if self.mt == 2 else ("v", 228 + 4 * tt)      # (K3b: v[228:235]; an MFMA's C and D share a FILE, not a register)

    def vaddr(self):
        return VADDR if self.mt == 2 else 236

    def tmp(self, k):
        return 224 + k

    def clobbers(self):
        out_regs = set() if self.mt == 2 else set(range(ACC, ACC + 16))      # (K3b: the outputs are tied to v[160:175], so they are operands, not clobbers)
        return ([f'"v{k}"' for k in range(self.last_v + 1) if k not in out_regs] + [f'"a{k}"' for k in range(self.last_a + 1)] +
                [f'"s{k}"' for k in range(40, 48)] + ['"vcc"', '"scc"', '"memory"'])


def generate(kinds, relu, dma_at=(1, 6, 11, 16, 21), epi_start=4, ko=0, prio=0, stagger=0, young_prio=0, mt=2):
    """ko (timing-only builds, results wrong; tools/k3a_lab.py): bit 0 no LDS-DMA, bit 1 no barrier, bit 2 no epilogue, bit 3 no
    fragment reads, bit 4 no waits for fragment reads, bit 5 every second fragment read only. prio: s_setprio for the whole statement.
    mt: ray tiles per wave (2 = K3a; 4 = K3b, which wants NW = 4)."""
    net = Net(kinds, relu)
    L = Lay(mt, NW)
    e = Emit()
    nchunks = len(net.chunks)
    # the fragments a wave really consumes, in order, with their chunk and place in it
    steps = []
    for c, (first, cnt) in enumerate(net.chunks):
        for q in range(cnt):
            f = net.frags[first + q]
            if f is not None:
                steps.append(dict(l=f[0], j=f[1], ks=f[2], tt=f[3], c=c, q=q))
    nsteps = len(steps)
    VA = L.vaddr()

    def in_which(l):       # the set layer l READS (layer 0 reads the features): 0 = A, 1 = B
        return 0 if l % 2 == 1 else 1

    def out_which(l):
        return 0 if l % 2 == 0 else 1

    def chunk_entry(c, first_of_pass=False):
        """In front of the first READ of chunk c: the wave's LDS-DMA pieces of it have landed, every wave says so, nobody reads
        chunk c - 2's slot any more (chunk c + 1 will be fetched into it)."""
        e.i(f"; ---- chunk {c}: its pieces have landed; all waves meet")
        e.i("s_waitcnt vmcnt(0)")
        if not ko & 2:
            e.i("s_barrier")
        if not first_of_pass:
            # s40 = slot of chunk c
            e.i(f"s_add_u32 s{S_T2}, s{S_RD}, {SLOT}")
            e.i(f"s_cmp_lt_u32 s{S_T2}, s{S_END}")
            e.i(f"s_cselect_b32 s{S_RD}, s{S_T2}, %[ring0]")
        # s41 = slot chunk c + 1 goes to (+ the wave's first piece)
        e.i(f"s_add_u32 s{S_T2}, s{S_RD}, {SLOT}")
        e.i(f"s_cmp_lt_u32 s{S_T2}, s{S_END}")
        e.i(f"s_cselect_b32 s{S_WR}, s{S_T2}, %[ring0]")
        e.i(f"s_add_u32 s{S_WR}, s{S_WR}, %[wavepiece]")
        e.i(f"v_add_u32 v{VA}, s{S_RD}, %[lane16]")

    def dma_piece(c_next, p):
        first, cnt = net.chunks[c_next % nchunks]
        if NW * p >= cnt or ko & 1:
            return
        e.i(f"; LDS-DMA: chunk {c_next % nchunks}{' of the next pass' if c_next >= nchunks else ''}, piece wave + {NW * p}")
        e.i(f"s_add_u32 s{S_T0}, s{S_SRC0}, {(first + NW * p) * 1024}")
        e.i(f"s_addc_u32 s{S_T1}, s{S_SRC1}, 0")
        e.i(f"s_add_u32 m0, s{S_WR}, {NW * p * 1024}")
        e.i("s_nop 0")
        e.i(f"global_load_lds_dwordx4 %[lane16], s[{S_T0}:{S_T1}]")

    def ring_read(si):
        s = steps[si]
        if ko & 8 or (ko & 32 and si % 2 == 1):
            return
        text = f"ds_read_b128 {L.reg(*L.ring(si % D))}, v{VA} offset:{s['q'] * 1024}"
        if ko & 16:
            e.i(text)
            return
        e.read(("ring", si), text)

    def bias_read(l, j):
        tiles = (0,) if net.kinds[l] in "LM" else (2 * j, 2 * j + 1)
        for t, tile in enumerate(tiles):
            e.read(("bias", l, j, t), f"ds_read_b128 {L.reg(*L.bias(t))}, %[biasv] offset:{4 * net.bias_base[l] + 64 * tile}")

    # K3b: the pairs of a pass in order, and the load of a pair's bias INTO its accumulators (C and D of an MFMA share a file)
    pairs = []
    for l, k in enumerate(net.kinds):
        pairs += [(l, 0)] if k in "LM" else [(l, j) for j in range(HT)]

    def bias_into_acc(pi, buf_):
        if pi >= len(pairs):
            return
        l, j = pairs[pi]
        lastl = net.kinds[l] in "LM"
        tiles = (0,) if lastl else (2 * j, 2 * j + 1)
        for t, tile in enumerate(tiles):
            for m in range(mt):
                dst = f"%[o{m}]" if lastl else L.reg(*L.acc(buf_, t, m))
                e.read(("biasacc", pi, t, m), f"ds_read_b128 {dst}, %[biasv] offset:{4 * net.bias_base[l] + 64 * tile}")

    def epilogue(l, j, buf):
        ins = []
        for m in range(mt):
            f, dst = L.act(out_which(l), j, m)
            # (an accumulator-file destination: convert into temporaries, ReLU there, move)
            tgt = [dst + k for k in range(4)] if f == "v" else [L.tmp(k) for k in range(4)]
            for t in (0, 1):
                a = L.acc(buf, t, m)[1]
                ins.append(f"v_cvt_pk_f16_f32 v{tgt[2 * t]}, v{a}, v{a + 1}")
                ins.append(f"v_cvt_pk_f16_f32 v{tgt[2 * t + 1]}, v{a + 2}, v{a + 3}")
            if net.relu[l]:
                for k in range(4):
                    ins.append(f"v_pk_max_f16 v{tgt[k]}, v{tgt[k]}, 0")
            if f == "a":
                for k in range(4):
                    ins.append(f"v_accvgpr_write_b32 a{dst + k}, v{tgt[k]}")
        return ins

    # ---------------------------------------------------------------- the pass
    e.i(f"; K3{'a' if mt == 2 else 'b'} pass body: layers {kinds}, hidden {32 * HT}, {nsteps} fragments, {nchunks} chunks, {NW} waves x {mt} ray tiles (generated by nif_asm_gen.py)")
    e.i(f"s_add_u32 s{S_END}, %[ring0], {RING_SLOTS * SLOT}")
    e.i(f"s_mov_b32 s{S_RD}, %[rd]")
    if stagger or young_prio:
        # vcc = "this is one of waves 4-7" (the younger wave of its SIMD: a workgroup's waves go to the SIMDs in cyclic order, so
        # waves w and w + 4 share one); nothing else in the statement touches vcc
        e.i("s_cmp_ge_u32 %[wavepiece], 4096")
        e.i("s_cselect_b64 vcc, -1, 0")
    if young_prio:
        # one static priority raise for the younger half, no flips (MI355X_MICROARCH.md, Two waves per SIMD, item 4)
        e.i("s_cbranch_vccz .Lk3a_oldhalf_%=")
        e.i(f"s_setprio {young_prio}")
        e.i(".Lk3a_oldhalf_%=:")
    if prio:
        e.i(f"s_setprio {prio}")
    e.i(f"s_add_u32 s{S_SRC0}, %[stream], %[wavepiece]")      # (64-bit operand: its low half; the high half follows)
    e.i(f"s_addc_u32 s{S_SRC1}, %[streamhi], 0")
    chunk_entry(0, first_of_pass=True)
    for si in range(min(D, nsteps)):
        assert steps[si]["c"] == 0
        ring_read(si)
    bias_read(0, 0)

    pending_epi = []            # VALU of the previous pair, dealt over the MFMA slots of the next pair from slot `epi_start` on
    pair_slot = 0               # MFMAs issued in the current pair
    buf = 0
    pair_i = -1                 # index into `pairs`
    epi_buf = None              # K3b: the buffer the pending epilogue reads (free for the pair after next once it is dealt)
    for si, s in enumerate(steps):
        l, j, ks, tt, c, q = s["l"], s["j"], s["ks"], s["tt"], s["c"], s["q"]
        kind = net.kinds[l]
        last = kind in "LM"
        KS = net.ks[l]
        if ks == 0 and tt == 0:
            pair_slot = 0
            pair_i += 1
        if not ko & 16:
            e.wait(("ring", si))
        if ks == 0:
            e.wait(("bias", l, j, tt))
        for m in range(mt):
            # B operand: the layer's input k-step, or a feature k-step
            act_steps = 0 if kind == "F" else HT
            if ks < act_steps:
                b = L.reg(*L.act(in_which(l), ks, m))
            else:
                b = f"%[f{ks - act_steps}{m}]"
            if last and mt == 2:
                d_ = f"%[o{m}]"
            elif last:
                # K3b: the final layer accumulates in the accumulator buffer whose turn it is - the statement's outputs are TIED to those
                # registers ("={v[160:163]}" ...: the register budget has no room for sixteen more operand registers)
                assert buf == 0, "K3b's outputs are tied to accumulator buffer 0: the network must have an even number of pairs before its final layer"
                d_ = L.reg(*L.acc(0, 0, m))
            else:
                d_ = L.reg(*L.acc(buf, tt, m))
            c_ = L.reg(*L.bias(tt)) if ks == 0 else d_
            e.i(f"v_mfma_f32_16x16x32_f16 {d_}, {L.reg(*L.ring(si % D))}, {b}, {c_}")
            pair_slot += 1
            # fillers behind this MFMA
            if m == mt - 1:
                nxt = si + D
                if nxt < nsteps:
                    if steps[nxt]["c"] != steps[nxt - 1]["c"]:
                        chunk_entry(steps[nxt]["c"])
                    ring_read(nxt)
            if pending_epi and pair_slot > epi_start:
                # the previous pair's epilogue is dealt over the FIRST HALF of this pair's MFMAs (all of them when the pair is
                # short): its values are the next layer's operands from that layer's k-step j on, and the final layer reads the
                # last pair's output in its last MFMAs
                total = (mt * KS) if last else (2 * mt * KS)
                until = total if total <= 4 * mt else total // 2
                per = max(1, -(-len(pending_epi) // max(1, until - pair_slot + 1)))
                for _ in range(min(per, len(pending_epi))):
                    e.i(pending_epi.pop(0))
        # the NEXT pair's bias, as soon as this pair's first MFMAs (which take the bias as C) have issued
        is_pair_first_done = (ks == 0 and (tt == 1 or last))
        if is_pair_first_done:
            if not last and j + 1 < HT:
                bias_read(l, j + 1)
            elif l + 1 < len(net.kinds) and not last:
                bias_read(l + 1, 0)
        # LDS-DMA pieces of the next chunk at fixed places of this chunk
        cnt_c = net.chunks[c][1]
        places = (dma_at if (NW == 8 or len(dma_at) == 10) else tuple(range(1, 31, 3))) if cnt_c == CH else (tuple(range(1, 2 * 5, 2)) if NW == 8 else tuple(range(10)))
        if stagger and cnt_c == CH:
            # the two waves of a SIMD issue their pieces `stagger` fragments apart: an LDS-DMA instruction holds its wave for tens of
            # cycles, and side by side both waves' MFMAs stop for them
            if q in places:
                p = places.index(q)
                e.i(f"s_cbranch_vccnz .Lk3a_s{si}_%=")
                dma_piece(c + 1, p)
                e.i(f".Lk3a_s{si}_%=:")
            if q - stagger in places:
                p = places.index(q - stagger)
                e.i(f"s_cbranch_vccz .Lk3a_t{si}_%=")
                dma_piece(c + 1, p)
                e.i(f".Lk3a_t{si}_%=:")
        elif q in places:
            p = places.index(q)
            dma_piece(c + 1, p)
        # end of a pair: its epilogue goes into the next pair's slots
        if not last and ks == KS - 1 and tt == 1:
            assert not pending_epi, "the previous pair's epilogue did not fit"
            pending_epi = [] if ko & 4 else epilogue(l, j, buf)
            buf ^= 1
    assert not pending_epi
    # the slot of the next pass's chunk 0 (= where this pass's last entry pointed the fetches)
    if prio or young_prio:
        e.i("s_setprio 0")
    e.i(f"s_sub_u32 %[rd], s{S_WR}, %[wavepiece]")
    e.i("s_nop 15")
    e.i("s_nop 15")
    generate.layout = L
    return net, e.render()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--kinds", default="FPPCPPL")
    ap.add_argument("--relu", default="")
    ap.add_argument("--dma-at", default="1,6,11,16,21")
    ap.add_argument("--epi-start", type=int, default=4)
    ap.add_argument("--ko", type=int, default=0, help="timing-only knock-outs (results wrong): 1 no LDS-DMA, 2 no barrier, 4 no epilogue, 8 no fragment reads")
    ap.add_argument("--prio", type=int, default=0)
    ap.add_argument("--ring", type=int, default=4, help="fragment ring depth")
    ap.add_argument("--waves", type=int, default=8, help="waves per workgroup (4: one wave per SIMD)")
    ap.add_argument("--tag", default="", help="suffix of the macro names written (nif_asm_kernel.hpp includes one body per workgroup shape)")
    ap.add_argument("--mt", type=int, default=2, help="ray tiles per wave: 2 (K3a, with --waves 8) or 4 (K3b, with --waves 4)")
    ap.add_argument("--stagger", type=int, default=0, help="waves 4-7 issue their LDS-DMA pieces this many fragments behind waves 0-3")
    ap.add_argument("--young-prio", type=int, default=0, help="static s_setprio for waves 4-7")
    a = ap.parse_args()
    set_ring_depth(a.ring)
    global NW
    NW = a.waves
    relu = [c == "1" for c in (a.relu or "1" * (len(a.kinds) - 1) + "0")]
    net, lines = generate(a.kinds, relu, tuple(int(x) for x in a.dma_at.replace(":", ",").split(",")), a.epi_start, a.ko, a.prio, a.stagger, a.young_prio, a.mt)
    T = a.tag
    with open(a.out, "w") as f:
        f.write(f"// generated by nif_asm_gen.py --kinds {a.kinds} --relu {''.join('1' if r else '0' for r in relu)} --waves {NW} --mt {a.mt}: do not edit\n")
        f.write(f"#define MI_NIF_ASM_KINDS{T} \"{a.kinds}\"\n#define MI_NIF_ASM_RELU{T} \"{''.join('1' if r else '0' for r in relu)}\"\n")
        f.write(f"#define MI_NIF_ASM_CHUNKS{T} {len(net.chunks)}\n#define MI_NIF_ASM_BIAS_FLOATS{T} {net.bias_floats}\n#define MI_NIF_ASM_WAVES{T} {NW}\n#define MI_NIF_ASM_MT{T} {a.mt}\n")
        f.write(f"#define MI_NIF_ASM_CLOBBERS{T} " + ", ".join(generate.layout.clobbers()) + "\n")
        f.write(f"#define MI_NIF_ASM_BODY{T} \\\n")
        for ln in lines:
            f.write(f'  "{ln}\\n\\t" \\\n')
        f.write('  ""\n')
    n_mfma = sum(1 for ln in lines if ln.startswith("v_mfma"))
    print(f"{a.out}: {len(lines)} lines, {n_mfma} MFMAs, {sum(1 for ln in lines if ln.startswith('ds_read'))} LDS reads, "
          f"{sum(1 for ln in lines if ln.startswith('global_load_lds'))} LDS-DMA pieces, {sum(1 for ln in lines if ln == 's_barrier')} barriers", file=sys.stderr)


if __name__ == "__main__":
    main()
