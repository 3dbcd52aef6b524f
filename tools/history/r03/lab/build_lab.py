"""Build EXPERIMENTAL variants of the library next to the product (tools/lab/_build/<name>/libomnivggt_hip.so).

The product sources are never touched: every experiment is a list of text substitutions applied to a COPY of csrc/ + include/,
compiled with the product's flags (omnivggt-official_amd/build.py). Unchanged translation units are compiled once
(_build/common/*.o) and linked into every variant. tools/lab/run_attn_lab.py times a variant's attention kernels against the
product's on the same box and checks the outputs. Nothing here is imported by the package, the tests or bench.py.

    python tools/lab/build_lab.py [name ...]         (default: all experiments)
    python tools/lab/build_lab.py --apply-to-product pipe_v2_q2   (promote an experiment: patches csrc/ in place; see apply_to_product)
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import importlib

B = importlib.import_module("omnivggt_official_amd.build")
import gen_pipe_body as G  # noqa: E402
import gen_pipe32_body as G32  # noqa: E402

OUT = os.path.join(HERE, "_build")

PIPE_PREFIX = r'''
  // ---- lab: order-pinned tile body (tools/lab/gen_pipe_body.py, schedule @NAME@) ----
#define PB_DSR(d, a, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(d) : "v"(a))
#define PB_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")")
#define PB_NOP() asm volatile("s_nop 0")
#define PB_EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define PB_CVT(d, a, b) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define PB_MFMA_NEW(d, a, b, c) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c))
#define PB_MFMA_ACC(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define PB_SPLIT(kt, qb) do { s[kt][qb][0] = t[kt][qb][0]; s[kt][qb][1] = t[kt][qb][1]; s[kt][qb][2] = t[kt][qb][2]; s[kt][qb][3] = t[kt][qb][3]; } while (0)
#define PB_PACK(u, qb) pf[u][qb] = u32x4{pw[u][qb][0], pw[u][qb][1], pw[u][qb][2], pw[u][qb][3]}
  constexpr bool PIPE = SM == 2 && (QB == 4 || (QB == 2 && @HAVE_Q2@)) && DMA > 0 && !VSUM && std::is_same<T, bf16_t>::value;
  auto pipe_tile = [&](int slot) {
    if constexpr (PIPE) {
      const uint32_t kb = lds_base + slot * SLOT_B + frag_row;
      const uint32_t ka0 = kb + coff0, ka1 = kb + coff1;
      const uint32_t va0 = kb + KT_B + (((0 + g) ^ sx) << 4), va1 = kb + KT_B + (((4 + g) ^ sx) << 4);
      u32x4 K[4][2], V[2][4], pf[2][QB];
      f32x4 t[4][QB];
      float s[4][QB][4];
      uint32_t pw[2][QB][4];
      if constexpr (QB == 4) {
@BODY@
      } else {
@BODY2@
      }
    }
  };
#undef PB_DSR
#undef PB_LGKM
#undef PB_NOP
#undef PB_EXP
#undef PB_CVT
#undef PB_MFMA_NEW
#undef PB_MFMA_ACC
#undef PB_SPLIT
#undef PB_PACK

  int since_barrier = 0;
'''

HEAD_OLD = '''  for (int j = 0; j < total_tiles; ++j) {
    const bool more = (j + 1) < total_tiles;'''
HEAD_NEW = '''  // lab: the loop body as a generic lambda, instantiated twice -- the order-pinned body for the leading FULL tiles of a
  // single-segment launch, the shipped body for the rest (the masked last tile; every tile of a multi-segment launch) -- so that
  // the hot loop holds ONE body (with both in one loop hipcc joins their register assignments with ~50 copies and spills O)
  auto tile_iter = [&](int j, auto use_pipe) {
    const bool more = (j + 1) < total_tiles;'''
LOOP_OLD = '''    f32x4 s[4][QB];
    qk_tile(kl, s, kv0 + BC > c_nk, kv0);          // the tail branch doubles as the scheduling fence (header)
    if constexpr (SM == 2) {'''
LOOP_NEW = '''    f32x4 s[4][QB];
    if constexpr (decltype(use_pipe)::value) {
      pipe_tile(buf);
    } else {
    qk_tile(kl, s, kv0 + BC > c_nk, kv0);          // the tail branch doubles as the scheduling fence (header)
    if constexpr (SM == 2) {'''
PV_OLD = '''    pv_step(vl, 0, s[0], s[1]);
    pv_step(vl, 1, s[2], s[3]);

    if (++ctile == c_ntiles) {'''
PV_NEW = '''    pv_step(vl, 0, s[0], s[1]);
    pv_step(vl, 1, s[2], s[3]);
    }

    if (++ctile == c_ntiles) {'''
END_OLD = '''  }
  if constexpr (DMA) __syncthreads();              // drain the tail transfers before the ring is reused (fallback pass) or the workgroup ends'''
END_NEW = '''  };
  int j_all = 0;
  if constexpr (PIPE) {
    // full tiles in front of the first masked one: single segment, tiles tile0 .. ; tile t is full iff (t + 1) * BC <= nk
    int n_full = p.nseg == 1 ? (int)(p.seg[0].nk / BC) - tile0 : 0;
    n_full = n_full < total_tiles ? n_full : total_tiles;
    for (; j_all < n_full; ++j_all) tile_iter(j_all, std::true_type{});
    asm volatile("s_nop 15\\n\\ts_nop 15");     // asm MFMA results -> VALU / builtin readers behind the loop (hipcc does not see the hazard)
  }
  for (; j_all < total_tiles; ++j_all) tile_iter(j_all, std::false_type{});
  if constexpr (DMA) __syncthreads();              // drain the tail transfers before the ring is reused (fallback pass) or the workgroup ends'''


def pipe_experiment(schedule, no_rowsum=False, schedule_q2=None):
    G.NO_ROWSUM = no_rowsum
    body = G.emit(G.generate(schedule))
    G.NO_ROWSUM = False
    body2 = G.emit(G.generate(schedule_q2)) if schedule_q2 else ""
    prefix = (PIPE_PREFIX.replace("@NAME@", schedule + (" / " + schedule_q2 if schedule_q2 else "")).replace("@BODY@", body)
              .replace("@BODY2@", body2).replace("@HAVE_Q2@", "true" if schedule_q2 else "false"))
    return [("ovg_attn16.h", "\n  int since_barrier = 0;\n", prefix), ("ovg_attn16.h", HEAD_OLD, HEAD_NEW), ("ovg_attn16.h", LOOP_OLD, LOOP_NEW),
            ("ovg_attn16.h", PV_OLD, PV_NEW), ("ovg_attn16.h", END_OLD, END_NEW)]


PRIO_OLD = '''  f32x4 o[QB][4], lacc[QB], negm[QB];
  if constexpr (MODE == 1) {'''


def prio_experiment(which):
    cond = {"young": "wave >= WAVES / 2", "old": "wave < WAVES / 2"}[which]
    return [("ovg_attn16.h", PRIO_OLD, '''  if (WAVES == 8 && (%s)) __builtin_amdgcn_s_setprio(1);   // lab: static priority for one half of the 8-wave workgroup
''' % cond + PRIO_OLD)]


SKEW_OLD = '''        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        __builtin_amdgcn_s_barrier();
        since_barrier = 0;'''


def skew_experiment(n):
    """8-wave kernel: the two waves of a SIMD leave every barrier together, i.e. IN phase (both in their MFMA group, then both in
    their VALU burst). Hold the second half of the workgroup back by n x 64 cycles behind each barrier so that the partners run in
    antiphase, like the free-running waves of the probe (and like the 4-wave kernel, whose SIMD partners belong to different
    workgroups and are never synchronised with each other)."""
    return [("ovg_attn16.h", SKEW_OLD, '''        asm volatile("s_waitcnt vmcnt(%%0)" ::"n"(NDMA) : "memory");
        __builtin_amdgcn_s_barrier();
        if (WAVES == 8 && wave_u >= 4) __builtin_amdgcn_s_sleep(%d);
        since_barrier = 0;''' % n)]


A32_INC_OLD = '''// variant (benchmark / test knob; numbers kept from the A/B logs under profiles/):'''
A32_BQ_OLD = '''  pl.bq = (v == 33 || v == 51 || v == 57 || v == 58 || v == 59) ? 512 : '''
A32_SLOTS_OLD = '''  const int slots = (v == 33 || v == 51 || v == 57 || v == 58 || v == 59) ? cus : 2 * cus;'''
A32_CASE_OLD = '''    case 59: return launch_attn16<T, 4, 8, 0, 2, false, 9>(p, pl, st);'''


def attn32_experiment(schedule):
    """The 32x32x16-MFMA speculative pass as a NEW file in the copy (ovg_attn32_lab.h) + its dispatch: variants 80 / 81 = order-pinned
    body with 512- / 256-row q tiles, 82 / 83 = the same kernels with the plain HIP body on every tile."""
    body = G32.emit(G32.generate(schedule))
    header = open(os.path.join(HERE, "ovg_attn32_lab.h.in")).read().replace("@NAME@", schedule).replace("@BODY@", body)
    return [("+ovg_attn32_lab.h", None, header),
            ("ovg_attn.hip", A32_INC_OLD, '#include "ovg_attn32_lab.h"\n' + A32_INC_OLD),
            ("ovg_attn.hip", A32_BQ_OLD, "  pl.bq = (v == 33 || v == 51 || v == 57 || v == 58 || v == 59 || v == 80 || v == 82) ? 512 : "),
            ("ovg_attn.hip", A32_SLOTS_OLD, "  const int slots = (v == 33 || v == 51 || v == 57 || v == 58 || v == 59 || v == 80 || v == 82) ? cus : 2 * cus;"),
            ("ovg_attn.hip", A32_CASE_OLD, A32_CASE_OLD + '''
    case 80: return launch_attn32<T, 8, 5, true>(p, pl, st);
    case 81: return launch_attn32<T, 4, 3, true>(p, pl, st);
    case 82: return launch_attn32<T, 8, 5, false>(p, pl, st);
    case 83: return launch_attn32<T, 4, 3, false>(p, pl, st);''')]


EXPERIMENTS = {
    "control": [],                                   # the product sources, rebuilt through the same path
    "prio_young": prio_experiment("young"),          # MI355X_MICROARCH.md, two waves per SIMD, item 4: s_setprio 1 for waves 4-7
    "prio_old": prio_experiment("old"),
    "pipe_v1": pipe_experiment("v1"),
    "pipe_v2": pipe_experiment("v2"),
    "pipe_v3": pipe_experiment("v3"),
    "pipe_v1_prio": pipe_experiment("v1") + prio_experiment("young"),
    "pipe_v2_norowsum": pipe_experiment("v2", no_rowsum=True),   # TIMING ONLY (garbage results): the price of the 8 row-sum MFMAs per tile
    "pipe_v2_q2": pipe_experiment("v2", schedule_q2="v2q2"),     # + the 128-row kernels (tail launch, frame / DINOv2 attention at 64 views)
    "pipe_v4": pipe_experiment("v4"),
    "pipe_v5": pipe_experiment("v5"),
    "attn32_w1": attn32_experiment("w1"),
    "attn32_w2": attn32_experiment("w2"),
    "attn32_x1": attn32_experiment("x1"),
    "attn32_x2": attn32_experiment("x2"),
    "skew1": skew_experiment(1),
    "skew2": skew_experiment(2),
    "pipe_v4_skew1": pipe_experiment("v4") + skew_experiment(1),
    "pipe_v4_skew2": pipe_experiment("v4") + skew_experiment(2),
    "pipe_v2_skew2": pipe_experiment("v2") + skew_experiment(2),
}


def hipcc_obj(csrc, src, obj):
    cmd = [B._hipcc(), *B.FLAGS, *B.EXTRA_FLAGS.get(src, []), "-c", os.path.join(csrc, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-6000:]))
    return obj


def build(name):
    subs = EXPERIMENTS[name]
    d = os.path.join(OUT, name)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    csrc = os.path.join(d, "pkg", "csrc")             # the sources include "../../include/omnivggt_hip.h"
    shutil.copytree(B.CSRC, csrc)
    os.makedirs(os.path.join(d, "include"))
    shutil.copy(os.path.join(ROOT, "include", "omnivggt_hip.h"), os.path.join(d, "include"))
    touched = set()
    for f, old, new in subs:
        if f.startswith("+"):                           # a new file of the copy
            open(os.path.join(csrc, f[1:]), "w").write(new)
            continue
        p = os.path.join(csrc, f)
        s = open(p).read()
        assert s.count(old) == 1, "experiment %s: anchor not found exactly once in %s: %r" % (name, f, old[:60])
        open(p, "w").write(s.replace(old, new))
        touched.add(f)
    # which translation units see a touched file? (headers are included by ovg_attn.hip / ovg_gemm.hip only; keep it simple)
    dirty = set()
    for src in B.SOURCES:
        text = open(os.path.join(csrc, src)).read()
        if src in touched or any(('"%s"' % h) in text for h in touched):
            dirty.add(src)
    common = os.path.join(OUT, "common")
    os.makedirs(common, exist_ok=True)

    def one(src):
        if src in dirty:
            return hipcc_obj(csrc, src, os.path.join(d, src.replace(".hip", ".o")))
        obj = os.path.join(common, src.replace(".hip", ".o"))
        if not os.path.exists(obj):
            hipcc_obj(B.CSRC, src, obj)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(one, B.SOURCES))
    so = os.path.join(d, "libomnivggt_hip.so")
    r = subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    for o in objs:
        if o.startswith(d):
            os.remove(o)                                # only the library travels to the GPU box; the patched sources stay for inspection
    return so, sorted(dirty)


def apply_to_product(name):
    """Write an experiment's substitutions into the PRODUCT sources (omnivggt-official_amd/csrc). For experiments that have earned it
    (pipe_v2: bit-identical outputs, +1.3 ... 5 %): afterwards rebuild (`__graft_entry__.build()`), re-take the traffic passes
    (tools/retake_traffic_r03.sh -> profiles/traffic.json; the attention source digest changes) and run the whole `-m gpu` suite."""
    for f, old, new in EXPERIMENTS[name]:
        assert not f.startswith("+"), "experiments that add lab-only files stay in the lab"
        p = os.path.join(B.CSRC, f)
        s = open(p).read()
        assert s.count(old) == 1, "anchor not found exactly once in %s: %r" % (f, old[:60])
        open(p, "w").write(s.replace(old, new))
        print("patched", os.path.relpath(p, ROOT))


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--apply-to-product":
        apply_to_product(sys.argv[2])
        sys.exit(0)
    names = sys.argv[1:] or list(EXPERIMENTS)
    for n in names:
        so, dirty = build(n)
        print("%-14s -> %s (recompiled: %s)" % (n, os.path.relpath(so, ROOT), ", ".join(dirty) or "nothing"))
