// Persistent 256 x 256 GEMM for the 16-bit modes (included by ovg_gemm.hip behind ovg_gemm256.h): the ping-pong LDS-DMA main loop
// of ovg_gemm256.h inside ONE workgroup per CU that walks a balanced share of the launch's work.
//
// Why (round 5, profiles/r05_gemm_timeline.txt -- per-workgroup wall-clock stamps of the r04 kernels): a K = 1024 tile spends 22-26 us
// in its main loop and 11-19 us OUTSIDE it -- 2.3-3.4 us until the first k-stage is visible, 0.5-1 us between workgroups, and an epilogue
// of 4 us (plain 16-bit store), 7-9 us (GELU / q-k norm + RoPE) or 12-15 us (f32 residual) during which the matrix pipe idles. The
// epilogues are NOT slow because the CUs run in lockstep (at 64 views ~57 of 256 CUs are in their epilogue at any time, and a forced
// start-up stagger changes nothing): a CU's store path moves ~16 B per clock, so the 128 KB (256 KB in f32) of a tile need 4 (8) us
// however they are issued, and the r04 epilogues ran their VALU work first and their stores afterwards. And at 8 views the launch
// itself quantises: 516 / 688 / 172 tiles on 256 CUs = 3 / 3 / 1 rounds where 2.02 / 2.69 / 0.67 would do.
//
// What this kernel does about it:
//   pieces     the launch's work is the list of its 256-row tiles in the grouped order of tile_coords(), each tile = 8 units of
//              32 rows (16 per wave group) plus a fixed cost of F units; workgroup w of P takes the slice [w C / P, (w + 1) C / P) of
//              that cost axis (C = tiles x (F + 8)). A slice boundary inside a tile splits it BY ROWS between two workgroups -- no
//              partial sums, no fix-up pass, every output row has one owner -- so a workgroup runs whole tiles plus at most two
//              partial ones ("pieces" of 32 mtc rows, mtc = 1..8 MFMA row blocks per wave group).
//   prefetch   the k-stage ring never drains between pieces: the last three L sections of a piece request the first three
//              k-stages of the next one (slots 0..2; nk % 4 == 0 keeps the slot phase), so the next main loop starts on landed data.
//   epilogues  run in chunks of 32 rows (16-bit outputs) or 16 rows (f32 residual) through 4 KB of slot 3 per wave: the stores of
//              chunk i drain while the VALU work of chunk i + 1 runs, the residual rows of chunk i + 1 are requested before chunk i
//              is stored, and slots 0..2 already hold the next piece.
//   X tiles    by buffer-addressed LDS-DMA: one descriptor per piece with num_records = the piece's rows, so rows past the piece
//              (or past M) read as zero without a per-lane clamp, and the per-lane offsets are loop constants.
// The main loop (ring, counted vmcnt, ping-pong wave groups, swizzle) and its RAW / WAR argument are those of ovg_gemm256.h; the
// differences are listed at run_piece().
#pragma once

namespace g256p {

using g256::BM2;
using g256::BN2;
using g256::ROWB;
using g256::SLOTS;
using g256::W_TILE;
using g256::X_TILE;
using g256::STAGE_B;
using g256::LDS_BYTES;
using g256::swz64;
using g256::lptr_t;

constexpr int UNIT_ROWS = 32, UPT = BM2 / UNIT_ROWS;      // work unit = 32 rows of one tile (16 per wave group); 8 per tile
constexpr int EPI_SLOT = 3;                                // ring slot the epilogues stage through (the slot of the piece's last k-stage)
constexpr int EPI_WAVE_BYTES = STAGE_B / 8;                // 4 KB per wave

struct Piece {        // wave-uniform
  int m0, n0;         // first row / column
  int mtc;            // 16-row MFMA blocks per wave group, 1..8: the piece covers rows [m0, m0 + 32 mtc)
  int row_end;        // min(M, m0 + 32 mtc)
};

// Slice of the cost axis -> pieces, in tile order. t is the walker's state (next tile to look at).
struct Walker {
  int c0, c1, F, M, mtiles, ntiles_gm, t, t_last;            // the cost axis fits 32 bits: tiles x (F + 8) x P < 2^31 for every shape ovg_linear accepts (M <= 2^30 / ... checked on the host)
  OVG_DEV void init(int w, int P, int M_, int N, int F_, int ntiles_gm_) {
    F = F_; M = M_; ntiles_gm = ntiles_gm_;
    mtiles = (M + BM2 - 1) / BM2;
    const int64_t T = (int64_t)mtiles * (N / BN2), C = T * (F + UPT);
    c0 = (int)((w * C) / P);
    c1 = (int)(((w + 1) * C) / P);
    t = c0 / (F + UPT);
    t_last = c1 > c0 ? (c1 - 1) / (F + UPT) : -1;
  }
  OVG_DEV bool next(Piece& pc) {
    while (t <= t_last) {
      const int ts = t * (F + UPT) + F;                      // cost coordinate of the tile's unit 0
      int lo = c0 - ts, hi = c1 - ts;
      lo = lo < 0 ? 0 : lo;
      hi = hi > UPT ? UPT : hi;
      int tm, tn;
      tile_coords(t, mtiles, ntiles_gm, tm, tn);
      ++t;
      const int rows = (M - tm * BM2) < BM2 ? (M - tm * BM2) : BM2;
      const int units = (rows + UNIT_ROWS - 1) / UNIT_ROWS;
      if (hi > units) hi = units;
      if (hi <= lo) continue;
      pc.m0 = tm * BM2 + lo * UNIT_ROWS;
      pc.n0 = tn * BN2;
      pc.mtc = hi - lo;
      const int e = pc.m0 + pc.mtc * UNIT_ROWS;
      pc.row_end = e < M ? e : M;
      return true;
    }
    return false;
  }
};

// The kernel's by-value parameter block, re-read from the kernarg segment (scalar loads) behind an opaque copy of the segment pointer: used
// once per piece in front of the epilogue, so that the ~40 SGPRs of parameters the epilogues need are not held (and spilled into VGPR lanes,
// with v_readlane in front of every DMA instruction of the main loop -- first build of this kernel) across the main loops.
template <typename P>
OVG_DEV P reload_params() {
  auto kp = __builtin_amdgcn_kernarg_segment_ptr();           // constant address space: the loads below are scalar
  asm volatile("" : "+s"(kp));
  P out;
  __builtin_memcpy(&out, (const void*)kp, sizeof(P));           // the address space is inferred back through the cast: s_load_dwordx*
  return out;
}

// (make_srd / buffer_dma16: ovg_common.h)

// Everything a workgroup keeps across its pieces: scalars only. The per-lane constants (DMA offsets, fragment offset) are recomputed at the
// start of every piece from an opaque copy of the thread id, so that they do not occupy VGPRs across the epilogues.
template <typename T>
struct Ctx {
  unsigned char* lds;
  uint32_t lds_base;                  // LDS byte address of lds (what M0 takes)
  const unsigned char* X;
  int ldxb, ldwb;                     // row strides in bytes
  i32x4 srd_w;
  int wave, wn, wm, nk;
  OVG_DEV void init(const T* Xp, int64_t ldx, const T* Wp, int64_t ldw, int N, int K, unsigned char* lds_) {
    lds = lds_;
    lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_);
    X = reinterpret_cast<const unsigned char*>(Xp);
    ldxb = (int)(ldx * 2); ldwb = (int)(ldw * 2);
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    wn = wave & 3; wm = wave >> 2;
    srd_w = make_srd(Wp, (uint32_t)((int64_t)N * ldwb));
    nk = (K * 2) / ROWB;
  }
  OVG_DEV i32x4 srd_x(const Piece& pc) const {     // rows past the piece read as zero
    return make_srd(X + (int64_t)pc.m0 * ldxb, (uint32_t)((int64_t)(pc.row_end - pc.m0) * ldxb));
  }
};

OVG_DEV void wait_vm(int n) {          // leave at most n k-stages (4 DMA instructions each) outstanding
  if (n >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// One piece's main loop. acc[nt][mt] = C[n = n0 + 64 wn + 16 nt + 4g + r][m = m0 + 16 mtc wm + 16 mt + (lane & 15)] for mt < mtc
// (SWAP: every 16 x 16 block transposed, see ovg_gemm256.h). Differences from g256::mainloop:
//  * `first` pieces stage k-stages 0..2 themselves; later pieces find them landed (requested by the previous piece, waited for -- vmcnt(0)
//    -- by every wave when ITS previous epilogue began, made visible by the barrier P below, which also says that every wave is done
//    with its epilogue's use of slot 3 before anybody requests k-stage 3 into it).
//    Their counted waits w(1), w(2) are skipped: the epilogue's own stores are still draining then, and a counted wait would wait for THEM.
//  * with `have_next` the last three L sections request the next piece's k-stages 0..2 and the counted waits keep two stages in flight
//    to the end.
//  * no trailing barrier pair: group 0 leaves behind b(2 nk - 1) and goes straight into its epilogue while group 1 runs its last M
//    section -- group 1's reads of slot 3 (L(nk - 1)) ended, lgkmcnt(0), in front of b(2 nk - 1) -- and group 1 drops its last barrier, so
//    both groups have executed P + 2 nk barriers.
//  * DYN: pieces with mtc < 8 read / multiply only their row blocks (scalar branches; full tiles keep the branch-free body).
template <typename T, bool SWAP, bool DYN>
OVG_DEV void run_piece(const Ctx<T>& cx, const Piece& pc, const bool first, const bool have_next, const Piece& nx, f32x4 (&acc)[4][8]) {
  const int nk = cx.nk, wave = cx.wave, mtc = DYN ? pc.mtc : 8;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));                              // opaque: what follows is recomputed per piece, not kept live across the epilogue
  const int lane = tid & 63, g = lane >> 4, lr = lane & 15;
  uint32_t voff_w[2], voff_x[2];                             // per-lane byte offsets of the two 16-row DMA pieces a wave stages per tile
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wave * 32 + i * 16 + (lane >> 2);
    const int ch = (lane & 3) ^ swz64(row);                  // source chunk for linear LDS slot (lane & 3)
    voff_w[i] = (uint32_t)(row * cx.ldwb + ch * 16);
    voff_x[i] = (uint32_t)(row * cx.ldxb + ch * 16);
  }
  const int frag_off = lr * ROWB + (g ^ swz64(lr)) * 16;
  const i32x4 sx_cur = cx.srd_x(pc), sx_nxt = cx.srd_x(nx);
  const uint32_t wsoff_cur = (uint32_t)pc.n0 * (uint32_t)cx.ldwb, wsoff_nxt = (uint32_t)nx.n0 * (uint32_t)cx.ldwb;
  auto stage = [&](int kt, bool next_piece) {
    const uint32_t wb = cx.lds_base + (kt & (SLOTS - 1)) * STAGE_B + wave * 32 * ROWB;   // wave-uniform destinations
    const uint32_t xb = wb + W_TILE;
    const uint32_t ko = (uint32_t)kt * ROWB;
    const uint32_t ws = (next_piece ? wsoff_nxt : wsoff_cur) + ko;
    const i32x4 sx = next_piece ? sx_nxt : sx_cur;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      buffer_dma16(wb + i * 16 * ROWB, voff_w[i], cx.srd_w, ws);
      buffer_dma16(xb + i * 16 * ROWB, voff_x[i], sx, ko);
    }
  };
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int w_off = cx.wn * 64 * ROWB + frag_off, x_off = W_TILE + cx.wm * 16 * mtc * ROWB + frag_off;
  u32x4 a[4], b[8];
  auto read_frags = [&](int kt) {
    const unsigned char* base = cx.lds + (kt & (SLOTS - 1)) * STAGE_B;
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = *reinterpret_cast<const u32x4*>(base + w_off + t * 16 * ROWB);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if constexpr (DYN) { if (t >= mtc) break; }
      b[t] = *reinterpret_cast<const u32x4*>(base + x_off + t * 16 * ROWB);
    }
  };
  auto mfmas = [&]() {
    __builtin_amdgcn_s_setprio(1);
    if constexpr (DYN) {
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        if (mt >= mtc) break;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          if constexpr (SWAP) TT<T>::mma(acc[nt][mt], b[mt], a[nt]);
          else TT<T>::mma(acc[nt][mt], a[nt], b[mt]);
        }
      }
    } else {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
          if constexpr (SWAP) TT<T>::mma(acc[nt][mt], b[mt], a[nt]);
          else TT<T>::mma(acc[nt][mt], a[nt], b[mt]);
        }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto request = [&](int t) {                       // the stage requested in L(t)
    if (t + 3 < nk) stage(t + 3, false);
    else if (have_next) stage(t + 3 - nk, true);
  };
  auto in_flight_after = [&](int t) {               // k-stages requested beyond stage t once L(min(t, ..)) has run
    if (have_next) return 2;
    const int last = (t + 2) < (nk - 1) ? (t + 2) : (nk - 1);
    return last - t;
  };

  if (first) {
    for (int s = 0; s < 3; ++s) stage(s, false);
    wait_vm(2);
  }
  __builtin_amdgcn_s_barrier();                      // P

  if (cx.wm == 0) {
    for (int t = 0; t < nk; ++t) {
      read_frags(t);                                 // L(t)
      request(t);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();                  // b(2t)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      mfmas();                                       // M(t)
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < nk && (first || t >= 2)) wait_vm(in_flight_after(t + 1));   // w(t+1); k-stages 1, 2 of a later piece have landed (see above)
      __builtin_amdgcn_s_barrier();                  // b(2t+1)
    }
  } else {
    __builtin_amdgcn_s_barrier();                    // b0: one barrier behind group 0
    for (int t = 0; t < nk; ++t) {
      read_frags(t);                                 // L(t)
      request(t);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < nk && (first || t >= 2)) wait_vm(in_flight_after(t + 1));   // w(t+1)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                  // b(2t+1)
      __builtin_amdgcn_sched_barrier(0);
      mfmas();                                       // M(t)
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < nk) __builtin_amdgcn_s_barrier();  // b(2t+2); the last one is dropped (see above)
    }
  }
  // the next piece's k-stages 0..2 (if any) must have landed before this wave's epilogue puts loads and stores of its own behind them
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------------------------------------------------------
// Chunked epilogues on a wave's 64 (n) x 16 mtc (m) block. img = the wave's 4 KB of ring slot 3; rows >= row_lim are not stored
// (row_lim = min(M, piece end, this wave group's 16 mtc rows)). Arithmetic identical to the staged epilogues of ovg_gemm.hip.
// ---------------------------------------------------------------------------------------------------------------------
// Addressing: every access of a chunk is base (SGPRs) + one per-lane 32-bit byte offset that is the same for all chunks, plus a wave-uniform
// row step folded into the scalar side -- 64-bit per-lane addresses for 8 chunks x 4 rows were what hipcc spilled in the first build of the
// residual form. The host sends a shape here only when (M - 1) * ld * sizeof + row bytes < 2^32 (persistent_legal_out).
template <typename T, int EPI>
OVG_DEV void epilogue16(const ovg_linear_params& p, f32x4 (&acc)[4][8], const int m_w0, const int n_w0, const int row_lim, const int mtc, unsigned char* img) {
  const int lane = threadIdx.x & 63, g = lane >> 4, lr = lane & 15;
  const int ncol = n_w0 + 4 * g;
  if (p.bias != nullptr) {                                    // the bias goes into the accumulators up front: nothing but acc is live across the chunks
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + ncol + nt * 16);
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) acc[nt][mt] += b;
    }
  }
  const int prow = lane >> 3, pch = lane & 7;                 // read-back shape: 8 rows x 8 chunks of 16 B = 8 whole lines per instruction
  const uint32_t ldyb = (uint32_t)p.ldy * (uint32_t)sizeof(T);
  const uint32_t off0 = (uint32_t)prow * ldyb + (uint32_t)(n_w0 + pch * 8) * (uint32_t)sizeof(T);
  unsigned char* ybase = static_cast<unsigned char*>(p.y) + (int64_t)m_w0 * ldyb;      // wave-uniform
  const int nrows = row_lim - m_w0;                           // valid rows of this wave's block (may be <= 0)
#pragma unroll
  for (int c = 0; c < 4; ++c) {                               // chunk c = row blocks 2c, 2c + 1 (32 rows x 128 B = the wave's 4 KB)
    if (2 * c < mtc) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int mt = 2 * c + h;
        f32x4 v[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) v[nt] = acc[nt][mt];
        if constexpr (EPI == OVG_EPI_GELU) {
          gelu_poly16(v);
          if constexpr (std::is_same<T, f16_t>::value) {      // f16 range guard (see linear_epilogue_impl)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
              for (int r = 0; r < 4; ++r) v[nt][r] = fminf(v[nt][r], 65504.0f);
          }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) stage16_put4<T>(img, h * 16 + lr, nt * 16 + 4 * g, v[nt][0], v[nt][1], v[nt][2], v[nt][3]);
      }
      u32x4 t[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = stage16_get(img, j * 8 + prow, pch);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r0 = c * 32 + j * 8;                        // wave-uniform first row of this instruction
        if (prow < nrows - r0) *reinterpret_cast<u32x4*>(ybase + (int64_t)r0 * ldyb + off0) = t[j];
      }
    }
  }
}

// f32 residual form: y = res + (gamma * acc + gamma * bias) [+ inject[m / period] on rows m % period == 0], one 16-row block (4 KB of f32)
// per chunk, the residual rows of the next block requested before this block's stores are issued. Camera-injection rows
// (omnivggt_aggregator.py:284-301; one row in 1374) are found per chunk with scalar arithmetic (period >= 128: at most one per chunk and one
// wrap per step) and added to the one row they belong to -- the non-persistent kernels send such waves through the register-form epilogue
// instead, whose 64 extra live registers this kernel cannot afford beside its loop state.
OVG_DEV void epilogue_res(const ovg_linear_params& p, f32x4 (&acc)[4][8], const int m_w0, const int n_w0, const int row_lim, const int mtc, unsigned char* img) {
  const int lane = threadIdx.x & 63, g = lane >> 4, lr = lane & 15;
  const int ncol = n_w0 + 4 * g;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {                            // acc <- gamma * acc + gamma * bias up front (the staged epilogue's expression): gamma / bias are dead before the chunks
    const f32x4 gam = *reinterpret_cast<const f32x4*>(p.gamma + ncol + nt * 16);
    f32x4 bias = p.bias != nullptr ? *reinterpret_cast<const f32x4*>(p.bias + ncol + nt * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
    bias = gam * bias;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) acc[nt][mt] = gam * acc[nt][mt] + bias;
  }
  const int prow = lane >> 4, pch = lane & 15;                // read-back shape: 4 rows x 16 chunks of 16 B (4 x 256 B) per instruction
  const uint32_t ldrb = (uint32_t)p.ldres * 4u, ldyb = (uint32_t)p.ldy * 4u;
  const uint32_t colb = (uint32_t)(n_w0 + pch * 4) * 4u;
  // residual offsets are absolute (from p.res): the clamp row below may lie in front of this wave's block
  const uint32_t roff0 = (uint32_t)(m_w0 + prow) * ldrb + colb, yoff0 = (uint32_t)prow * ldyb + colb;
  const unsigned char* rbase = reinterpret_cast<const unsigned char*>(p.res);                          // wave-uniform
  unsigned char* ybase = static_cast<unsigned char*>(p.y) + (int64_t)m_w0 * ldyb;
  const int nrows = row_lim - m_w0;                           // valid rows of this wave's block (may be <= 0)
  const uint32_t roff_safe = (uint32_t)(row_lim - 1) * ldrb + colb;   // dead rows load a live row (the last valid one) instead of branching
  const bool has_inj = p.inject != nullptr;
  const int period = has_inj ? (int)p.inj_period : 1 << 30;
  int vw0 = 0, rem0 = 0;                                     // m_w0 = vw0 * period + rem0 (wave-uniform)
  if (has_inj) {
    const FastDiv d(period);
    d.divmod(m_w0, vw0, rem0);
    vw0 = __builtin_amdgcn_readfirstlane(vw0);
    rem0 = __builtin_amdgcn_readfirstlane(rem0);
  }
  f32x4 res[2][4];
  auto fetch = [&](int mt, f32x4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r0 = mt * 16 + i * 4;                         // wave-uniform
      const uint32_t off = prow < nrows - r0 ? roff0 + (uint32_t)r0 * ldrb : roff_safe;
      r[i] = *reinterpret_cast<const f32x4*>(rbase + off);
    }
  };
  fetch(0, res[0]);
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    if (mt < mtc) {
      if (mt + 1 < mtc) fetch(mt + 1, res[(mt + 1) & 1]);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) stage32_put4(img, lr, nt * 16 + 4 * g, acc[nt][mt]);
      f32x4 t[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) t[i] = stage32_get(img, i * 4 + prow, pch);
#pragma unroll
      for (int i = 0; i < 4; ++i) t[i] = res[mt & 1][i] + t[i];
      if (has_inj) {                                          // wave-uniform
        int rem = rem0 + mt * 16, vw = vw0;
        if (rem >= period) { rem -= period; ++vw; }
        const int r_inj = rem == 0 ? 0 : period - rem;        // local row of the chunk's injection row, if < 16
        if (r_inj < 16) {
          if (rem != 0) ++vw;
          const f32x4 iv = *reinterpret_cast<const f32x4*>(p.inject + (int64_t)vw * p.N + n_w0 + pch * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (i * 4 + prow == r_inj) t[i] += iv;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r0 = mt * 16 + i * 4;
        if (prow < nrows - r0) *reinterpret_cast<f32x4*>(ybase + (int64_t)r0 * ldyb + yoff0) = t[i];
      }
    }
  }
}

// q / k tiles of the QKV projection: bias + per-head LayerNorm(64) + 2-D RoPE + q scale (the arithmetic of qk_rows / qk_epilogue in ovg_gemm.hip,
// 16-bit outputs), the 64-wide head rows staged 32 tokens at a time and stored as whole 128-byte lines.
template <typename T, bool NORM, bool ROPE>
OVG_DEV void qk_rows_chunked(const f32x4 (&acc)[4][8], const float* __restrict__ nw_p, const float* __restrict__ nb_p,
                             const float* __restrict__ rope_cos, const float* __restrict__ rope_sin, T* __restrict__ out, const int64_t npad,
                             const int m_w0, const int row_lim, const int mtc, const int seq, const int h, const int tokens_per_view, const int n_special,
                             const int grid_w, const float qk_eps, const float scale, unsigned char* img) {
  const int lane = threadIdx.x & 63, g = lane >> 4, lr = lane & 15;
  const FastDiv div_seq(seq), div_tpv(ROPE ? tokens_per_view : 1), div_gw(ROPE ? grid_w : 1);
  const int last = row_lim - 1;
  float nw[16], nb[16];
  if constexpr (NORM) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(nw_p + nt * 16 + 4 * g);
      const f32x4 b = *reinterpret_cast<const f32x4*>(nb_p + nt * 16 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) { nw[nt * 4 + r] = a[r]; nb[nt * 4 + r] = b[r]; }
    }
  }
  const int prow = lane >> 3, pch = lane & 7;                 // 8 tokens x 8 chunks of 16 B per instruction: 8 whole 128-byte head rows
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    if (mt < mtc) {
      const int m = m_w0 + mt * 16 + lr;
      const int mm = m < last ? m : last;
      f32x4 cy, sy, cx, sxn;
      if constexpr (ROPE) {
        int vw, t, py, px;
        div_tpv.divmod(mm, vw, t);
        const int pp = t - n_special;
        div_gw.divmod(pp >= 0 ? pp : 0, py, px);
        py = pp >= 0 ? py + 1 : 0;
        px = pp >= 0 ? px + 1 : 0;
        cy = *reinterpret_cast<const f32x4*>(rope_cos + py * 16 + 4 * g);
        sy = *reinterpret_cast<const f32x4*>(rope_sin + py * 16 + 4 * g);
        cx = *reinterpret_cast<const f32x4*>(rope_cos + px * 16 + 4 * g);
        sxn = *reinterpret_cast<const f32x4*>(rope_sin + px * 16 + 4 * g);
      }
      float v[16];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[nt * 4 + r] = acc[nt][mt][r];      // bias already added by the caller
      if constexpr (NORM) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += v[i];
        s = quad16_sum(s);
        const float mean = s * (1.0f / 64.0f);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const float d = v[i] - mean; q += d * d; }
        q = quad16_sum(q);
        const float rstd = __builtin_amdgcn_rsqf(q * (1.0f / 64.0f) + qk_eps);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = (v[i] - mean) * rstd * nw[i] + nb[i];
      }
      if constexpr (ROPE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a0 = v[r], a1 = v[4 + r];          // features j, j+16 of the y half
          v[r] = a0 * cy[r] - a1 * sy[r];
          v[4 + r] = a1 * cy[r] + a0 * sy[r];
          const float b0 = v[8 + r], b1 = v[12 + r];     // features j, j+16 of the x half
          v[8 + r] = b0 * cx[r] - b1 * sxn[r];
          v[12 + r] = b1 * cx[r] + b0 * sxn[r];
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] *= scale;          // 1.0 for k (exact), q_scale for q
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) stage16_put4<T>(img, (mt & 1) * 16 + lr, nt * 16 + 4 * g, v[nt * 4], v[nt * 4 + 1], v[nt * 4 + 2], v[nt * 4 + 3]);
      if ((mt & 1) || mt + 1 >= mtc) {                     // 32 tokens staged (or the block is the piece's last): store them
        const int rows0 = m_w0 + (mt & ~1) * 16;
        u32x4 t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = stage16_get(img, j * 8 + prow, pch);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int mr = rows0 + j * 8 + prow;
          int bidx, n;
          div_seq.divmod(mr < last ? mr : last, bidx, n);
          if (mr < row_lim && (j < 2 || (mt & 1))) *reinterpret_cast<u32x4*>(out + (((int64_t)bidx * OVG_H + h) * npad + n) * OVG_D + pch * 8) = t[j];
        }
      }
    }
  }
}

template <typename T>
OVG_DEV void qk_epilogue_chunked(const ovg_qkv_params& p, f32x4 (&acc)[4][8], const int m_w0, const int ncol0_, const int row_lim, const int mtc,
                                 const float* rope_c, const float* rope_s, unsigned char* img) {
  const int ncol0 = __builtin_amdgcn_readfirstlane(ncol0_);   // wave-uniform: keep the q/k dispatch scalar
  const int lane = threadIdx.x & 63, g = lane >> 4;
  const int which = ncol0 / OVG_C;                      // 0 q, 1 k
  const int h = (ncol0 % OVG_C) / OVG_D;                // head of this wave's 64 columns
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + ncol0 + nt * 16 + 4 * g);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) acc[nt][mt] += b;
  }
  const float* nw_p = which == 0 ? p.qn_w : p.kn_w;
  const float* nb_p = which == 0 ? p.qn_b : p.kn_b;
  T* out = static_cast<T*>(which == 0 ? p.q : p.k);
  const int64_t npad = which == 0 ? p.nq_pad : p.nk_pad;
  const float scale = which == 0 ? p.q_scale : 1.0f;
  const int tpv = (int)p.tokens_per_view;
#define OVG_QK_ROWS_P(NORM, ROPE) qk_rows_chunked<T, NORM, ROPE>(acc, nw_p, nb_p, rope_c, rope_s, out, npad, m_w0, row_lim, mtc, (int)p.seq, h, \
                                                                   tpv, p.n_special, p.grid_w, p.qk_eps, scale, img)
  if (p.qk_norm) { if (p.rope) OVG_QK_ROWS_P(true, true); else OVG_QK_ROWS_P(true, false); }
  else { if (p.rope) OVG_QK_ROWS_P(false, true); else OVG_QK_ROWS_P(false, false); }
#undef OVG_QK_ROWS_P
}

}  // namespace g256p
