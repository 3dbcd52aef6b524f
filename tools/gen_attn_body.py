"""Generator of the ORDER-PINNED tile body of the bf16 speculative attention kernel (csrc/ovg_attn16.h, run_tiles<SM = 2>).

    python tools/gen_attn_body.py --write     regenerate csrc/ovg_attn16_body_q4.inc (schedule v2) and _q2.inc (schedule v2q2)
    python tools/gen_attn_body.py --check     exit 1 if the committed .inc files differ from what the generator emits
    python tools/gen_attn_body.py <schedule>  print one schedule's body (the round-3 lab schedules are kept for reference)

Left to hipcc, the instruction order of a tile clusters: [32 QK^T MFMAs][32 v_exp + 16 v_cvt_pk][40 PV MFMAs with the other 32
exps in front]; every attempt to make hipcc interleave (sched_group_barrier, sched_barrier fences, source-level pipelining) lost
to register spills or to its own clustering. Here the order is pinned instead: every MFMA, v_exp_f32, v_cvt_pk_bf16_f32,
ds_read_b128 and s_waitcnt of a full (unmasked) tile is ONE `asm volatile` statement -- hipcc never reorders volatile asm
statements against each other, but still allocates the registers (checked: no copies, exps run in place on the MFMA result
registers, the converts write straight into the P fragment tuples). What hipcc no longer does for us, and this generator
therefore checks on the emitted stream (verify()):
  * MFMA result -> VALU read needs 8 wait states on gfx950 (4-pass XDL op); we demand >= 8 instructions in between;
  * v_exp (transcendental) result -> v_cvt_pk read: >= 2 instructions in between;
  * v_cvt_pk result -> MFMA B operand: >= 2 instructions in between;
  * asm ds_reads are invisible to hipcc's s_waitcnt insertion: the generator tracks the in-order LDS return queue and emits
    the counted s_waitcnt lgkmcnt(n) in front of the first consumer of every fragment.
Same arithmetic, same operation order per accumulator as the compiler-scheduled body (which still runs the masked last tile and
every tile of a multi-segment launch) -> bit-identical results. Measured in the round-3 lab (profiles/r03_attn_lab_*.txt,
history: tools/history/r03/lab): schedule v2 (one exp per MFMA, evenly spread) +1.3...2.3 % at 64 views, +5 % at 8 views;
v2q2 the same order for the 128-row kernels (tail launch, frame-local / DINOv2 attention).
"""
import sys

QB = 4
NO_ROWSUM = False


def mfma_stream():
    """The 72 MFMAs of a tile in issue order: A(kt) = QK^T of key sub-tile kt (k-step 0 for the 4 q blocks, then k-step 1),
    P(u) = row-sum + PV of the 32-key step u, q block by q block."""
    ops = []
    for kt in range(4):
        for qb in range(QB):
            ops.append(("QK1", kt, qb))
        for qb in range(QB):
            ops.append(("QK2", kt, qb))
    for u in range(2):
        for qb in range(QB):
            ops.append(("L", u, qb))
            for dt in range(4):
                ops.append(("PV", u, qb, dt))
    return ops


# LDS fragment reads: ("K", kt, h) -> K[kt][h], ("V", u, dt) -> V[u][dt]
TOP_READS = [("K", 0, 0), ("K", 0, 1), ("K", 1, 0), ("K", 1, 1)]

# fragment reads, placed as late as the LDS latency (~64-128 cycles = 4-8 MFMA slots) allows: every fragment register is
# live for a short time only -- the body runs at the 256-VGPR limit (128 persistent: O 64, row sums 16, anchors 16, Q 32)
EARLY_READS = {1: [("K", 2, 0)], 3: [("K", 2, 1)], 9: [("K", 3, 0)], 11: [("K", 3, 1)],
               17: [("V", 0, 0)], 19: [("V", 0, 1)], 21: [("V", 0, 2)], 23: [("V", 0, 3)],
               33: [("V", 1, 0)], 35: [("V", 1, 1)], 37: [("V", 1, 2)], 39: [("V", 1, 3)]}
LATE_READS = {7: [("K", 2, 0)], 9: [("K", 2, 1)], 15: [("K", 3, 0)], 17: [("K", 3, 1)],
              25: [("V", 0, 0)], 27: [("V", 0, 1)], 29: [("V", 0, 2)], 31: [("V", 0, 3)],
              44: [("V", 1, 0)], 46: [("V", 1, 1)], 48: [("V", 1, 2)], 50: [("V", 1, 3)]}

SCHEDULES = {
    # name: (exp quota per MFMA slot, cvt quota per MFMA slot, {slot: [reads issued after that MFMA]})
    # v1: exps two per MFMA from the first legal slot on (A1: E0, A2: E1, A3: E2, P0: E3), converts as soon as legal
    "v1": dict(exp=lambda m: 2 if 8 <= m < 32 else (1 if m >= 32 else 0), cvt=lambda m: 1 if m >= 17 else 0,
               reads=LATE_READS),
    # v2: one exp per MFMA from slot 8 to the end (evenly spread), converts as soon as legal
    "v2": dict(exp=lambda m: 1 if m >= 8 else 0, cvt=lambda m: 1 if m >= 17 else 0,
               reads=LATE_READS),
    # v3: three exps per two MFMAs (1.5), between v1 and v2
    "v3": dict(exp=lambda m: (2 if m % 2 == 0 else 1) if m >= 8 else 0, cvt=lambda m: 1 if m >= 17 else 0,
               reads=LATE_READS),
    # v4: CLUSTERS. tools/probes/mfma_valu_mix (profiles/r03_probe_mfma_valu_mix.txt): with two waves per SIMD the attention mix runs at
    # 39.5 cycles per MFMA pair when 8 MFMAs alternate with their ~100 cycles of VALU work, against 45-49 for fine interleaving
    # (1, 2 or 4 MFMAs per group) and 45-47 for the 16 / 32 / 72-MFMA clusters hipcc produces: MFMA groups of 8-16 with bursts of
    # ~100-130 VALU cycles between them, so that the partner wave's MFMA group fits under this wave's burst and vice versa.
    #   [A0 A1] E0 [A2] E1 [A3] C0 + E2/4 [P0 q0 q1] E2 rest + E3/4 [P0 q2 q3] E3 rest + C1 q0 q1 [P1 q0 q1] C1 q2 q3 [P1 q2 q3]
    "v4": dict(exp=lambda m: {15: 16, 23: 16, 31: 4, 41: 16, 51: 12}.get(m, 0), cvt=lambda m: {31: 16, 51: 8, 61: 8}.get(m, 0),
               reads={3: [("K", 2, 0)], 5: [("K", 2, 1)], 11: [("K", 3, 0)], 13: [("K", 3, 1)],
                      25: [("V", 0, 0)], 27: [("V", 0, 1)], 29: [("V", 0, 2)], 31: [("V", 0, 3)],
                      44: [("V", 1, 0)], 46: [("V", 1, 1)], 48: [("V", 1, 2)], 50: [("V", 1, 3)]}),
    # v5: smaller clusters: MFMA groups of 8 / 5, bursts of ~64 cycles
    # v2q2: the v2 idea for the 128-row kernels (QB = 2: 36 MFMAs, 32 exps, 16 converts per tile; the tail launch of the 64-view plan and the
    # frame-local / DINOv2 attention at 64 views): one exp per MFMA as soon as the hazard distance allows
    "v2q2": dict(qb=2, exp=lambda m: 1 if m >= 5 else 0, cvt=lambda m: 1 if m >= 9 else 0,
                 reads={1: [("K", 2, 0)], 3: [("K", 2, 1)], 5: [("K", 3, 0)], 7: [("K", 3, 1)],
                        9: [("V", 0, 0)], 11: [("V", 0, 1)], 13: [("V", 0, 2)], 15: [("V", 0, 3)],
                        19: [("V", 1, 0)], 21: [("V", 1, 1)], 23: [("V", 1, 2)], 25: [("V", 1, 3)]}),
    "v5": dict(exp=lambda m: {15: 8, 19: 8, 23: 8, 27: 8, 31: 4, 36: 6, 41: 6, 46: 6, 51: 6, 56: 4}.get(m, 0),
               cvt=lambda m: {31: 8, 36: 4, 41: 4, 46: 4, 51: 4, 56: 4, 61: 4}.get(m, 0),
               reads={3: [("K", 2, 0)], 5: [("K", 2, 1)], 11: [("K", 3, 0)], 13: [("K", 3, 1)],
                      25: [("V", 0, 0)], 27: [("V", 0, 1)], 29: [("V", 0, 2)], 31: [("V", 0, 3)],
                      44: [("V", 1, 0)], 46: [("V", 1, 1)], 48: [("V", 1, 2)], 50: [("V", 1, 3)]}),
}


def exp_order(name):
    """Order in which the 64 exps are issued: (kt, qb, r). v1 / v3: key sub-tile by key sub-tile. v2: E0, E1 whole, then
    q block by q block for E2 / E3 (P(1) consumes them q block by q block)."""
    if name == "v2":
        o = [(kt, qb, r) for kt in (0, 1) for qb in range(QB) for r in range(4)]
        o += [(kt, qb, r) for qb in range(QB) for kt in (2, 3) for r in range(4)]
        return o
    return [(kt, qb, r) for kt in range(4) for qb in range(QB) for r in range(4)]


def cvt_order():
    return [(u, qb, w) for u in range(2) for qb in range(QB) for w in range(4)]


def cvt_sources(u, qb, w):
    kt = 2 * u + (w >> 1)
    r0 = 2 * (w & 1)
    return [(kt, qb, r0), (kt, qb, r0 + 1)]


def generate(name):
    global QB
    sch = SCHEDULES[name]
    QB = sch.get("qb", 4)
    mf = mfma_stream()
    stream = []          # (kind, payload) in issue order
    lds_queue = []       # outstanding reads, oldest first
    ready_frag = set()   # fragments already waited for

    def need(frag):
        if frag in ready_frag:
            return
        assert frag in lds_queue, "fragment %r used before it was read" % (frag,)
        idx = lds_queue.index(frag)
        younger = len(lds_queue) - idx - 1
        stream.append(("WAIT", younger))
        for f in lds_queue[: idx + 1]:
            ready_frag.add(f)
        del lds_queue[: idx + 1]

    def read(frag):
        stream.append(("READ", frag))
        lds_queue.append(frag)

    for f in TOP_READS:
        read(f)
    exps, cvts = exp_order(name), cvt_order()
    pos = {}             # op -> index in stream where it was issued

    def issued_at(key):
        return pos.get(key)

    def exp_ready(e):
        kt, qb, r = e
        p = issued_at(("QK2", kt, qb))
        return p is not None and len(stream) - p - 1 >= 8

    def cvt_ready(c):
        for s in cvt_sources(*c):
            p = issued_at(("E",) + s)
            if p is None or len(stream) - p - 1 < 2:
                return False
        return True

    for m, op in enumerate(mf):
        if op[0] in ("QK1", "QK2"):
            need(("K", op[1], 0 if op[0] == "QK1" else 1))
        if op[0] == "PV":
            need(("V", op[1], op[3]))
        if op[0] in ("L", "PV"):
            for w in range(4):
                p = issued_at(("C", op[1], op[2], w))
                assert p is not None, "P fragment (%d,%d) not converted before MFMA slot %d" % (op[1], op[2], m)
                while len(stream) - p - 1 < 2:
                    stream.append(("NOP", 0))
        pos[op] = len(stream)
        stream.append(("MFMA", op))
        for f in sch["reads"].get(m, []):
            read(f)
        ne, nc = sch["exp"](m), sch["cvt"](m)
        for _ in range(ne):
            if exps and exp_ready(exps[0]):
                e = exps.pop(0)
                pos[("E",) + e] = len(stream)
                stream.append(("EXP", e))
        for _ in range(nc):
            if cvts and cvt_ready(cvts[0]):
                c = cvts.pop(0)
                pos[("C",) + c] = len(stream)
                stream.append(("CVT", c))
        # a P(u) MFMA group must not start before its converts exist: flush what the quota left behind
        if m + 1 < len(mf) and mf[m + 1][0] in ("L", "PV"):
            u, qb = mf[m + 1][1], mf[m + 1][2]
            want = [("C", u, qb, w) for w in range(4)]
            guard = 0
            while any(k not in pos for k in want):
                guard += 1
                assert guard < 400, "cannot satisfy the converts of P(%d, %d)" % (u, qb)
                if cvts and cvt_ready(cvts[0]):
                    c = cvts.pop(0)
                    pos[("C",) + c] = len(stream)
                    stream.append(("CVT", c))
                elif exps and exp_ready(exps[0]):
                    e = exps.pop(0)
                    pos[("E",) + e] = len(stream)
                    stream.append(("EXP", e))
                else:
                    stream.append(("NOP", 0))
    assert not exps and not cvts, "schedule left %d exps / %d converts unissued" % (len(exps), len(cvts))
    assert not lds_queue, "reads never consumed: %r" % lds_queue
    verify(stream)
    return stream


def verify(stream):
    """Independent check of the hazard distances and of the LDS wait accounting on the final stream."""
    pos = {}
    outstanding = []
    landed = set()
    for i, (kind, x) in enumerate(stream):
        if kind == "READ":
            outstanding.append(x)
        elif kind == "WAIT":
            keep = x
            while len(outstanding) > keep:
                landed.add(outstanding.pop(0))
        elif kind == "MFMA":
            if x[0] == "QK1":
                assert ("K", x[1], 0) in landed, (i, x)
            elif x[0] == "QK2":
                assert ("K", x[1], 1) in landed, (i, x)
                assert i - pos[("QK1", x[1], x[2])] >= 1
            elif x[0] == "PV":
                assert ("V", x[1], x[3]) in landed, (i, x)
            if x[0] in ("L", "PV"):
                for w in range(4):
                    assert i - pos[("C", x[1], x[2], w)] - 1 >= 2, (i, x)
            pos[x] = i
        elif kind == "EXP":
            assert i - pos[("QK2", x[0], x[1])] - 1 >= 8, (i, x)
            pos[("E",) + x] = i
        elif kind == "CVT":
            for s in cvt_sources(*x):
                assert i - pos[("E",) + s] - 1 >= 2, (i, x)
            pos[("C",) + x] = i
    assert not outstanding


def emit(stream):
    out = []
    for kind, x in stream:
        if kind == "READ":
            if x[0] == "K":
                out.append("PB_DSR(K[%d][%d], ka%d, %d);" % (x[1], x[2], x[2], x[1] * 2048))
            else:
                out.append("PB_DSR(V[%d][%d], va%d, %d);" % (x[1], x[2], x[1], x[2] * 2048))
        elif kind == "WAIT":
            out.append("PB_LGKM(%d);" % x)
        elif kind == "NOP":
            out.append("PB_NOP();")
        elif kind == "EXP":
            out.append("PB_EXP(s[%d][%d][%d]);" % x)
        elif kind == "CVT":
            u, qb, w = x
            (k0, _, r0), (_, _, r1) = cvt_sources(u, qb, w)
            out.append("PB_CVT(pw[%d][%d][%d], s[%d][%d][%d], s[%d][%d][%d]);" % (u, qb, w, k0, qb, r0, k0, qb, r1))
        elif kind == "MFMA":
            if x[0] == "QK1":
                out.append("PB_MFMA_NEW(t[%d][%d], K[%d][0], qf[%d][0], negm[%d]);" % (x[1], x[2], x[1], x[2], x[2]))
            elif x[0] == "QK2":
                kt, qb = x[1], x[2]
                out.append("PB_MFMA_ACC(t[%d][%d], K[%d][1], qf[%d][1]); PB_SPLIT(%d, %d);" % (kt, qb, kt, qb, kt, qb))
            elif x[0] == "L":
                u, qb = x[1], x[2]
                if NO_ROWSUM:      # timing-only experiment: what would the launch cost WITHOUT the 8 row-sum MFMAs (results are garbage: l stays 0)
                    out.append("PB_PACK(%d, %d);" % (u, qb))
                else:
                    out.append("PB_PACK(%d, %d); PB_MFMA_ACC(lacc[%d], ones, pf[%d][%d]);" % (u, qb, qb, u, qb))
            else:
                u, qb, dt = x[1], x[2], x[3]
                out.append("PB_MFMA_ACC(o[%d][%d], V[%d][%d], pf[%d][%d]);" % (qb, dt, u, dt, u, qb))
    return "\n".join("      " + l for l in out)


def summary(stream):
    n = {}
    for kind, _ in stream:
        n[kind] = n.get(kind, 0) + 1
    return n


PRODUCT = {"ovg_attn16_body_q4.inc": "v2", "ovg_attn16_body_q2.inc": "v2q2"}


def product_text(schedule):
    st = generate(schedule)
    n = summary(st)
    head = ("// GENERATED by tools/gen_attn_body.py (schedule %s: %d MFMA, %d v_exp_f32, %d v_cvt_pk, %d ds_read_b128, %d counted waits, %d s_nop) -- do not edit;\n"
            "// hazard distances and LDS wait counts are verified by the generator (tests/test_attn_body_generator.py re-checks this file against it).\n"
            % (schedule, n.get("MFMA", 0), n.get("EXP", 0), n.get("CVT", 0), n.get("READ", 0), n.get("WAIT", 0), n.get("NOP", 0)))
    return head + emit(st) + "\n"


def product_paths():
    import os
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "omnivggt-official_amd", "csrc")
    return {os.path.join(csrc, f): sch for f, sch in PRODUCT.items()}


if __name__ == "__main__":
    arg = sys.argv[1] if len(sys.argv) > 1 else "--check"
    if arg in ("--write", "--check"):
        bad = 0
        for path, sch in product_paths().items():
            text = product_text(sch)
            if arg == "--write":
                open(path, "w").write(text)
                print("wrote", path)
            elif (not __import__("os").path.exists(path)) or open(path).read() != text:
                print("STALE:", path)
                bad = 1
        sys.exit(bad)
    st = generate(arg)
    sys.stderr.write("schedule %s: %r\n" % (arg, summary(st)))
    print(emit(st))
