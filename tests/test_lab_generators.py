"""The attention-lab body generators (tools/lab/): every schedule must produce a stream that passes its own hazard / wait-count
verification and contains exactly one tile's worth of work. (CPU only: the generated bodies are measured on the GPU through
tools/lab/run_attn_lab.py, not through this suite.)"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "lab"))


def _count(stream):
    n = {}
    for kind, _ in stream:
        n[kind] = n.get(kind, 0) + 1
    return n


@pytest.mark.parametrize("schedule", ["v1", "v2", "v3", "v4", "v5"])
def test_pinned_16x16x32_body_schedules_verify(schedule):
    import gen_pipe_body as G
    st = G.generate(schedule)                      # generate() runs verify(): MFMA -> exp >= 8, exp -> cvt >= 2, cvt -> MFMA >= 2, counted lgkmcnt
    n = _count(st)
    assert (n["MFMA"], n["EXP"], n["CVT"], n["READ"]) == (72, 64, 32, 16)
    text = G.emit(st)
    assert text.count("PB_MFMA_NEW") == 16 and text.count("PB_MFMA_ACC") == 56 and text.count("PB_DSR") == 16
    # per accumulator the operation order of the shipped body: u = 0 before u = 1 for every (q block, feature tile)
    for qb in range(4):
        for dt in range(4):
            a, b = text.index("PB_MFMA_ACC(o[%d][%d], V[0][%d]" % (qb, dt, dt)), text.index("PB_MFMA_ACC(o[%d][%d], V[1][%d]" % (qb, dt, dt))
            assert a < b


@pytest.mark.parametrize("schedule", ["w1", "w2", "x1", "x2"])
def test_pinned_32x32x16_body_schedules_verify(schedule):
    import gen_pipe32_body as G
    st = G.generate(schedule)
    n = _count(st)
    assert (n["MFMA"], n["EXP"], n["CVT"], n["READ"]) == (40, 64, 32, 16)
    # every P fragment = registers {4 a1 + b, 4 (a1 + 2) + b} of its S^T block (chunk 2 a1 + h of the V^T rows as the QKV kernel stores them)
    for a1 in range(2):
        regs = sorted(r for w in range(4) for (_, _, r) in G.cvt_sources(0, a1, 0, w))
        assert regs == [4 * a1 + b for b in range(4)] + [4 * a1 + 8 + b for b in range(4)]


def test_lab_experiments_still_anchor_in_the_product_sources():
    """tools/lab/build_lab.py patches COPIES of csrc/ by exact text substitution: every anchor of every experiment must occur exactly
    once in the current product sources, in the order the experiment applies them -- otherwise the lab (and the promotion of an
    experiment with --apply-to-product) has silently rotted after an edit of the kernels."""
    import build_lab as BL
    csrc = BL.B.CSRC
    for name, subs in BL.EXPERIMENTS.items():
        text = {}
        for f, old, new in subs:
            if f.startswith("+"):
                continue
            if f not in text:
                text[f] = open(os.path.join(csrc, f)).read()
            assert text[f].count(old) == 1, (name, f, old[:70])
            text[f] = text[f].replace(old, new)
