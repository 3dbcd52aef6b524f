"""Order-pinned tile body of the 32x32x16-MFMA attention formulation (lab experiment, round 3).

Why 32x32x16: tools/probes/mfma_valu_mix (profiles/r03_probe_mfma_valu_mix.txt) shows that a SIMD issues ONE instruction at a time
across its two waves and that an MFMA costs ~9.4 issue cycles whatever its shape: with 16x16x32 MFMAs the attention mix (0.9 exp +
0.45 cvt + 0.35 other per MFMA) needs 41-45 cycles per MFMA pair against 32 of matrix pipe (the shipped kernel sits AT that bound);
with 32x32x16 MFMAs (twice the FLOP per issue slot) the same mix runs at 69 cycles per pair against 66 of matrix pipe = 95 %.

Tile = 64 keys x 64 q rows per wave = 2 key blocks (kb) x 2 q blocks (qb) of 32; head dim 64 = 4 k-steps (ks) of 16.
  S^T[kb][qb] (+)= K[kb][ks] Q[qb][ks]                         16 MFMAs   (A = K rows from LDS, B = Q, registers)
  lane (q = lane % 32, h = lane / 32) then holds S^T rows (keys) 32 kb + 8 a + 4 h + b in register i = 4 a + b
  P fragment of the PV k-step (kb, a1) = registers {4 a1 + b, 4 (a1 + 2) + b}: keys 8 a1 + 4 h + b and 16 + 8 a1 + 4 h + b of block kb
  = exactly chunk g = 2 a1 + h of the V^T rows as the QKV kernel stores them (vt_pos16): no change of any other kernel
  l[qb]      += 1 P[kb][a1][qb]                                 8 MFMAs
  O^T[db][qb] += V^T[kb][a1][db] P[kb][a1][qb]                  16 MFMAs   (db = block of 32 features)
No anchor: P = exp2(s) directly (bf16 P and f32 sums have the f32 exponent range; the verified-fallback check of the kernel
covers overflow / underflow exactly as it covers a bad anchor today) -- the first QK^T MFMA of a block takes the constant 0 as C.

    python tools/lab/gen_pipe32_body.py [schedule] > body.inc
"""
import sys


def mfma_stream(order="kb_ks_qb"):
    """QK^T issue order: kb_ks_qb = the two q blocks alternate (same accumulator every 2nd MFMA); kb_qb_ks = the 4 k-steps of one
    accumulator back to back; ks_kb_qb = four accumulators round-robin (same accumulator every 4th MFMA, exps start after all of QK^T)."""
    ops = []
    if order == "kb_ks_qb":
        ops = [("QK", kb, ks, qb) for kb in range(2) for ks in range(4) for qb in range(2)]
    elif order == "kb_qb_ks":
        ops = [("QK", kb, ks, qb) for kb in range(2) for qb in range(2) for ks in range(4)]
    else:
        ops = [("QK", kb, ks, qb) for ks in range(4) for kb in range(2) for qb in range(2)]
    for kb in range(2):
        for a1 in range(2):
            for qb in range(2):
                ops.append(("L", kb, a1, qb))
            for db in range(2):
                for qb in range(2):
                    ops.append(("PV", kb, a1, db, qb))
    return ops


TOP_READS = [("K", 0, 0), ("K", 0, 1), ("K", 0, 2), ("K", 0, 3)]
READS = {1: [("K", 1, 0)], 3: [("K", 1, 1)], 5: [("K", 1, 2)], 7: [("K", 1, 3)],
         11: [("V", 0, 0, 0)], 13: [("V", 0, 0, 1)], 17: [("V", 0, 1, 0)], 19: [("V", 0, 1, 1)],
         23: [("V", 1, 0, 0)], 25: [("V", 1, 0, 1)], 29: [("V", 1, 1, 0)], 31: [("V", 1, 1, 1)]}

READS_RR = {0: [("K", 0, 2)], 1: [("K", 1, 2)], 4: [("K", 0, 3)], 5: [("K", 1, 3)],
            11: [("V", 0, 0, 0)], 13: [("V", 0, 0, 1)], 17: [("V", 0, 1, 0)], 19: [("V", 0, 1, 1)],
            23: [("V", 1, 0, 0)], 25: [("V", 1, 0, 1)], 29: [("V", 1, 1, 0)], 31: [("V", 1, 1, 1)]}

SCHEDULES = {
    # exps per MFMA slot / converts per MFMA slot (issued only when their operands are legal, see generate)
    "w1": dict(exp=lambda m: 3 if m >= 8 else 0, cvt=lambda m: 2 if m >= 10 else 0),
    "w2": dict(exp=lambda m: 2 if m >= 8 else 0, cvt=lambda m: 1 if m >= 10 else 0),
    "x1": dict(exp=lambda m: 3 if m >= 8 else 0, cvt=lambda m: 2 if m >= 10 else 0, order="kb_qb_ks"),
    "x2": dict(exp=lambda m: 3 if m >= 8 else 0, cvt=lambda m: 2 if m >= 10 else 0, order="ks_kb_qb",
               top=[("K", 0, 0), ("K", 1, 0), ("K", 0, 1), ("K", 1, 1)], reads=READS_RR),
}

MFMA_TO_VALU = 12     # 8-pass XDL op -> VALU read of its result: 12 wait states on gfx950


def exp_order():
    o = []
    for kb in range(2):
        for a1 in range(2):
            for qb in range(2):
                o += [(kb, qb, 4 * a1 + b) for b in range(4)] + [(kb, qb, 4 * a1 + 8 + b) for b in range(4)]
    return o


def cvt_order():
    return [(kb, a1, qb, w) for kb in range(2) for a1 in range(2) for qb in range(2) for w in range(4)]


def cvt_sources(kb, a1, qb, w):
    i0 = 4 * a1 + 8 * (w >> 1) + 2 * (w & 1)
    return [(kb, qb, i0), (kb, qb, i0 + 1)]


def generate(name):
    sch = SCHEDULES[name]
    mf = mfma_stream(sch.get("order", "kb_ks_qb"))
    top_reads, reads = sch.get("top", TOP_READS), sch.get("reads", READS)
    stream, lds_queue, ready = [], [], set()
    pos = {}

    def need(frag):
        if frag in ready:
            return
        assert frag in lds_queue, "fragment %r used before it was read" % (frag,)
        idx = lds_queue.index(frag)
        stream.append(("WAIT", len(lds_queue) - idx - 1))
        for f in lds_queue[: idx + 1]:
            ready.add(f)
        del lds_queue[: idx + 1]

    def read(frag):
        stream.append(("READ", frag))
        lds_queue.append(frag)

    for f in top_reads:
        read(f)
    exps, cvts = exp_order(), cvt_order()

    def exp_ready(e):
        p = pos.get(("QK", e[0], 3, e[1]))
        return p is not None and len(stream) - p - 1 >= MFMA_TO_VALU

    def cvt_ready(c):
        for s in cvt_sources(*c):
            p = pos.get(("E",) + s)
            if p is None or len(stream) - p - 1 < 2:
                return False
        return True

    def pop_exp():
        e = exps.pop(0)
        pos[("E",) + e] = len(stream)
        stream.append(("EXP", e))

    def pop_cvt():
        c = cvts.pop(0)
        pos[("C",) + c] = len(stream)
        stream.append(("CVT", c))

    for m, op in enumerate(mf):
        if op[0] == "QK":
            need(("K", op[1], op[2]))
        if op[0] == "PV":
            need(("V", op[1], op[2], op[3]))
        if op[0] in ("L", "PV"):
            kb, a1, qb = op[1], op[2], op[-1]
            want = [("C", kb, a1, qb, w) for w in range(4)]
            guard = 0
            while any(k not in pos for k in want):
                guard += 1
                assert guard < 400, "cannot satisfy the converts of P(%d, %d, %d)" % (kb, a1, qb)
                if cvts and cvt_ready(cvts[0]):
                    pop_cvt()
                elif exps and exp_ready(exps[0]):
                    pop_exp()
                else:
                    stream.append(("NOP", 0))
            while any(len(stream) - pos[k] - 1 < 2 for k in want):
                stream.append(("NOP", 0))
        pos[op] = len(stream)
        stream.append(("MFMA", op))
        for f in reads.get(m, []):
            read(f)
        for _ in range(sch["exp"](m)):
            if exps and exp_ready(exps[0]):
                pop_exp()
        for _ in range(sch["cvt"](m)):
            if cvts and cvt_ready(cvts[0]):
                pop_cvt()
    assert not exps and not cvts and not lds_queue, (len(exps), len(cvts), lds_queue)
    verify(stream)
    return stream


def verify(stream):
    pos, outstanding, landed = {}, [], set()
    for i, (kind, x) in enumerate(stream):
        if kind == "READ":
            outstanding.append(x)
        elif kind == "WAIT":
            while len(outstanding) > x:
                landed.add(outstanding.pop(0))
        elif kind == "MFMA":
            if x[0] == "QK":
                assert ("K", x[1], x[2]) in landed, (i, x)
                if x[2] > 0:
                    assert ("QK", x[1], x[2] - 1, x[3]) in pos
            elif x[0] == "PV":
                assert ("V", x[1], x[2], x[3]) in landed, (i, x)
            if x[0] in ("L", "PV"):
                for w in range(4):
                    assert i - pos[("C", x[1], x[2], x[-1], w)] - 1 >= 2, (i, x)
            pos[x] = i
        elif kind == "EXP":
            assert i - pos[("QK", x[0], 3, x[1])] - 1 >= MFMA_TO_VALU, (i, x)
            pos[("E",) + x] = i
        elif kind == "CVT":
            for s in cvt_sources(*x):
                assert i - pos[("E",) + s] - 1 >= 2, (i, x)
            pos[("C",) + x] = i
    assert not outstanding
    n = sum(1 for k, _ in stream if k == "EXP"), sum(1 for k, _ in stream if k == "CVT"), sum(1 for k, _ in stream if k == "MFMA")
    assert n == (64, 32, 40), n


def emit(stream):
    out = []
    for kind, x in stream:
        if kind == "READ":
            if x[0] == "K":
                out.append("PC_DSR(K[%d][%d], ka[%d], %d);" % (x[1], x[2], x[2], x[1] * 4096))
            else:
                out.append("PC_DSR(V[%d][%d][%d], ka[%d], %d);" % (x[1], x[2], x[3], 2 * x[1] + x[2], 8192 + x[3] * 4096))
        elif kind == "WAIT":
            out.append("PC_LGKM(%d);" % x)
        elif kind == "NOP":
            out.append("PC_NOP();")
        elif kind == "EXP":
            out.append("PC_EXP(s[%d][%d][%d]);" % x)
        elif kind == "CVT":
            kb, a1, qb, w = x
            (_, _, i0), (_, _, i1) = cvt_sources(kb, a1, qb, w)
            out.append("PC_CVT(pw[%d][%d][%d][%d], s[%d][%d][%d], s[%d][%d][%d]);" % (kb, a1, qb, w, kb, qb, i0, kb, qb, i1))
        elif kind == "MFMA":
            if x[0] == "QK":
                kb, ks, qb = x[1], x[2], x[3]
                if ks == 0:
                    out.append("PC_MFMA_NEW0(t[%d][%d], K[%d][%d], qf[%d][%d]);" % (kb, qb, kb, ks, qb, ks))
                else:
                    out.append("PC_MFMA_ACC(t[%d][%d], K[%d][%d], qf[%d][%d]);%s" % (kb, qb, kb, ks, qb, ks, " PC_SPLIT(%d, %d);" % (kb, qb) if ks == 3 else ""))
            elif x[0] == "L":
                kb, a1, qb = x[1], x[2], x[3]
                out.append("PC_PACK(%d, %d, %d); PC_MFMA_ACC(lacc[%d], ones, pf[%d][%d][%d]);" % (kb, a1, qb, qb, kb, a1, qb))
            else:
                kb, a1, db, qb = x[1], x[2], x[3], x[4]
                out.append("PC_MFMA_ACC(o[%d][%d], V[%d][%d][%d], pf[%d][%d][%d]);" % (db, qb, kb, a1, db, kb, a1, qb))
    return "\n".join("      " + l for l in out)


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "w1"
    st = generate(name)
    kinds = {}
    for k, _ in st:
        kinds[k] = kinds.get(k, 0) + 1
    sys.stderr.write("schedule %s: %r\n" % (name, kinds))
    print(emit(st))
