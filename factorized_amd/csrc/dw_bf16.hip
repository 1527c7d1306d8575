// Weight gradients of the LSTMs (and of the decoders' fc1) on a bf16-RESIDENT plan: ONE pass over the gate gradients.
//
// Reference: what loss.backward() accumulates into weight_ih / weight_hh / bias_ih / bias_hh of every nn.LSTMCell and into
// decoderLSTM.fc1 (mfm_model.py:40-91 unrolled over T).  With A_t = the gate pre-activation gradients the BPTT left behind:
//     dW_ih = sum_t A_t^T x_t        dW_hh = sum_{t>=1} A_t^T h_{t-1}        db_ih = db_hh = sum_t A_t^T 1
//     dW_fc = sum_t dxhat_t^T h_t    db_fc = sum_t dxhat_t^T 1
// i.e. per item ONE product  C[M, N] += A^T [seg_0 | seg_1]  over all T*B rows, plus the column sums of A.
//
// Why a kernel of its own (round 3): on a bf16-resident plan dA, h and (a padded copy of) x sit in HBM as bf16, ROW-major,
// and the reduction runs over the rows -- both MFMA operands are "k-strided".  The grouped GEMM (gemm_bf16.hip) has to
// transpose every tile through registers on its way into LDS and re-reads dA once per consumer (1.27 GB for ~0.4 GB of
// operands at B=2048, 369 us: profiles/r02_roofline_table_l_bf16.txt).  Here nothing is transposed or converted in registers:
//   * global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): the [32 rows][columns] slabs land in LDS in memory order,
//     no staging registers, three chunks in flight per workgroup (counted vmcnt, raw s_barrier);
//   * LDS -> MFMA fragments by ds_read_b64_tr_b16, the gfx950 transposing read: lane (bi, q) asks for 4 consecutive
//     columns of row 4q + bi/4 and receives column bi of rows 4q .. 4q+3 (scripts/micro/glds_tr_probe.hip); two reads
//     (rows +0, +16) make the 8-deep k slice of v_mfma_f32_16x16x32_bf16 -- A and B fragments use the same k assignment,
//     and the MFMA sums over all of k;
//   * a workgroup owns 96 columns of A and ALL N columns of the right-hand side for a range of rows, so A is read once and
//     the right-hand side once per 96 A-columns -- from the SAME XCD's L2, because the M-tiles of one row range are
//     consecutive workgroups of one XCD;
//   * out-of-range DMA lanes write zeros (probed), so ragged row ranges and the t = 0 rows of h_{t-1} need no branches:
//     their offsets are simply outside the buffer resource.
// 512 threads = 8 waves as 2 (MF A fragments each) x 4 (N fragments, strided, <= NFW each); tile shapes (MF, NFW): (3, 9) 96 x <= 576
// columns, (4, 8) / (4, 9) 128 x <= 512 / 576, (8, 4) 256 x <= 256 -- 27 to 36 accumulator tiles per wave.
//
// Round 5 (profiles/r05_dw_stream_study.txt): a clock on the chunk loop's phases (scripts/dwb_chunk_timeline.sh) showed that the
// launch is not memory bound -- a chunk costs ~0.9 k cycles of waits and barriers plus ~0.5 k per DMA instruction, and its fragment
// reads keep the LDS busy ~2.5 x as long as its MFMAs keep the matrix pipe.  Hence: the launcher prices a chunk per DMA instruction
// (narrow items had been starved), picks the tile shape PER ITEM (dw_stream_mixed_kernel runs 128- and 256-column bodies in one
// launch; a right-hand side of more than 256 columns is split into column parts whose workgroups stay neighbours on one XCD),
// gives narrow items 64 or 128 rows per chunk, staggers the DMA issue of the two waves of a SIMD, requests fragments one group ahead
// and spreads the bias sums over the waves.  B = 2048: 137 + 20 us -> 100 + 16 us.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>

#include "internal.h"

namespace mfm {

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int DWB_THREADS = 512;
constexpr int DWB_MF = 3;                       // A fragments per wave
constexpr int DWB_MT = 2 * 16 * DWB_MF;         // A columns per workgroup: 96
constexpr int DWB_NFW = 9;                      // N fragments per wave: N (padded) <= 16 * 4 * 9 = 576
constexpr int DWB_KC = 32;                      // rows per chunk = one MFMA k-block
constexpr int DWB_MIN_STAGES = 3, DWB_MAX_STAGES = 12;    // chunks of LDS an item rotates through (DwbItem::stages)
constexpr int DWB_MAX_WAIT = 24;                           // largest counted vmcnt wait: (stages - 2) x instructions per chunk
constexpr int DWB_MAXNI = 6;                    // LDS-DMA instructions per thread and chunk
#ifndef DWB_STAGGER
#define DWB_STAGGER 1                           // bf16 form: half of the waves issue the next chunk's DMA after their MFMAs (see the chunk loop)
#endif

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* lds_base, int off, int row16_bytes) {
  // two transposing reads: rows [0,16) and [16,32) of the chunk, 4 k each for this lane
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_base + off));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_base + off + row16_bytes));
  const bf16x4 a = __builtin_bit_cast(bf16x4, lo), b = __builtin_bit_cast(bf16x4, hi);
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// F32 = true (round 3, fp32 plans at large T*B; OPT-IN, see the status note at the launcher): the same one-pass structure on
// fp32 buffers and v_mfma_f32_16x16x4_f32 -- 16-row chunks (four 4-deep k-steps), 4 floats per DMA piece, fragments by plain
// ds_read_b32 (lane (bi, q) reads element [row 4 ks + q][column bi] of the memory-order slab: no transposition problem at 4
// bytes per element).  The column groups of odd slab rows are rotated by 16 floats on their way in (the DMA source address
// is free, the LDS side is lane-linear), so the two rows a 32-lane half of a fragment read touches fall on different banks.
// MF: A fragments per wave (the workgroup owns MT = 32 MF columns of A), NFW: N fragments per wave (N <= 64 NFW).  (3, 9) is the
// general form; (4, 8) -- 128-column M-tiles, N <= 512 -- re-reads the right-hand side once per 128 instead of once per 96
// columns of A and has no half-empty last tile for M = 128 / 256 / 512 (the launcher picks it when every item fits).
// ---- which tile a workgroup owns.  XCD-aware order: workgroup b runs on XCD b % 8; the tiles of one (item, row range) -- every
// M-tile of every column part -- become consecutive workgroups of ONE XCD, so the operands they share are fetched from HBM once
// and then hit that L2.
struct DwbTile { int item, mt, sp, local; };
__device__ __forceinline__ DwbTile dwb_tile(const DwbLaunch& L) {
  int v;
  {
    const int nb = (int)gridDim.x, x = (int)blockIdx.x % 8, j = (int)blockIdx.x / 8;
    const int per = nb / 8, rem = nb % 8;
    v = x * per + (x < rem ? x : rem) + j;
  }
  int head = 0;                                    // (entries with part > 0 carry tile_begin = INT_MAX: never picked here)
#pragma unroll
  for (int i = 1; i < MFM_DWB_MAXI; ++i) head = (i < L.n_items && v >= L.it[i].tile_begin) ? i : head;
  const DwbItem& H = L.it[head];
  const int vv = v - H.tile_begin, per = H.parts * H.m_tiles;
  DwbTile t;
  t.sp = vv / per;
  const int rem = vv - t.sp * per, part = rem / H.m_tiles;
  t.mt = rem - part * H.m_tiles;
  // (integer division runs on the vector ALU: back into scalar registers, or everything derived from the entry -- the LDS
  // addresses that go into M0 -- would be vector values)
  t.sp = __builtin_amdgcn_readfirstlane(t.sp);
  t.mt = __builtin_amdgcn_readfirstlane(t.mt);
  t.item = __builtin_amdgcn_readfirstlane(head + part);
  t.local = t.sp * H.m_tiles + t.mt;               // index of the partial tile inside its entry (slab form)
  return t;
}

template <bool F32, int MF, int NFW>
__device__ __forceinline__ void dw_stream_body(const DwbLaunch& L, const DwbItem& I, const int mt, const int sp, const int local) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  constexpr int MT = 2 * 16 * MF;
  constexpr int ES = F32 ? 4 : 2;                 // bytes per operand element
  constexpr int EPP = 16 / ES;                    // elements per 16-byte DMA piece
  constexpr int KC = F32 ? 16 : DWB_KC;           // rows per chunk
#ifdef MFM_DWB_STAMP
  const long long st_kernel = __builtin_readcyclecounter();
#endif
  const int m0 = mt * MT;
  const int r_begin = sp * I.rows_per_split, r_end = min(L.rows, r_begin + I.rows_per_split);
  // rows per chunk: KC x (k-blocks per chunk).  A chunk costs ~0.9 k cycles of waits and barriers plus ~0.5 k per DMA instruction
  // whatever its width, so narrow items (a decoder's [32 rows][128 + 32 columns] is 10 KB) take 64 or 128 rows per chunk.
  const int kpc = F32 ? 1 : I.kpc;
  const int KR = KC * kpc;
  const int kr_shift = F32 ? 4 : (kpc == 4 ? 7 : kpc == 2 ? 6 : 5);          // KR is a power of two: no integer divisions by it
  const int n_chunks = (r_end - r_begin + KR - 1) >> kr_shift;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bi = lane & 15, q = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;

  // ---- chunk image in LDS, in DMA piece order (16 bytes per piece, lane-linear): A [32][96] | seg0 [32][n0] | seg1 [32][n1]
  const int n0 = I.seg[0].ncols, n1 = (I.nseg > 1) ? I.seg[1].ncols : 0;
  const int pa = KR * MT / EPP, p0 = KR * n0 / EPP, p1 = KR * n1 / EPP;            // pieces; each a multiple of 64
  const int P = pa + p0 + p1;
  const int NI = (P + DWB_THREADS - 1) / DWB_THREADS;
  const int stage_bytes = NI * DWB_THREADS * 16;
  const int NF = (n0 + n1) >> 4;
  // bf16 images whose rows are a multiple of 256 bytes (128 / 256 columns) would put the 16 rows of a transposing fragment
  // read on the same banks: their 16-byte pieces are rotated by 2 (row & 7) positions on the way in (the DMA source address
  // is free), and the fragment reads below undo it
  // (such a width has >= 16 pieces per row, so 2 (row & 7) <= 14 needs no reduction.  The set-up below avoids run-time integer
  // divisions where it can -- ~40 of them, ~40 VALU instructions each, made the prologue 10 k cycles: a third of a workgroup's
  // life at T*B = 2560 rows)
  auto bf_rot = [](int width, int row) { return ((width & 127) == 0) ? 2 * (row & 7) : 0; };
  auto wrap = [](int x, int n) { return x >= n ? x - n : x; };               // x mod n for 0 <= x < 2 n

  // ---- per-thread DMA plan: piece p = i * 512 + tid -> (image, row of the chunk, 8-column group).  Plain global addresses
  // (global_load_dwordx4 ... lds), one 64-bit pointer per piece advanced by a per-piece stride each chunk; a piece that
  // must read as zero -- rows of A at or beyond r_end, rows of h_{t-1} before the first time step, rows past the end of a
  // segment, columns past a row's end, idle lanes of the last instruction -- is pointed at a 16-byte block of zeros.
  const unsigned char* src[DWB_MAXNI];
  // a piece is valid while its row, row0 + (rows per chunk) x chunk, lies in [0, hi): kept as the chunk range [first, first + count)
  // packed into one register, vld = first | count << 16 (a row range has < 65536 chunks)
  int inc[DWB_MAXNI];
  unsigned vld[DWB_MAXNI];
  auto chunk_range = [&](int row0, int hi) -> unsigned {
    // chunks c with 0 <= row0 + KR c < hi
    const int first = row0 >= 0 ? 0 : (-row0 + KR - 1) >> kr_shift;
    const int end = hi > row0 ? (hi - row0 + KR - 1) >> kr_shift : 0;       // first chunk with row >= hi
    const int count = end > first ? min(end - first, 65535) : 0;
    return (unsigned)min(first, 65535) | ((unsigned)count << 16);
  };
  const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(L.zeros);
  const unsigned magic0 = (1u << 20) / (unsigned)max(n0 / EPP, 1) + 1u, magic1 = (1u << 20) / (unsigned)max(n1 / EPP, 1) + 1u;
#pragma unroll
  for (int i = 0; i < DWB_MAXNI; ++i) {
    const int p = i * DWB_THREADS + tid;
    src[i] = zsrc; inc[i] = 0; vld[i] = 0;                      // never valid
    // F32: the column groups of odd slab rows are rotated by 4 pieces (16 floats): LDS position c of row `row` holds source
    // group (c + 4 (row & 1)) mod groups-per-row (bank spreading for the b32 fragment reads, see the kernel comment)
    const unsigned char* abase = reinterpret_cast<const unsigned char*>(I.a);
    if (p < pa) {
      constexpr int GPR = MT / EPP;
      const int row = p / GPR, cpos = p % GPR;
      const int cg = F32 ? (cpos + 4 * (row & 1)) % GPR : wrap(cpos + bf_rot(MT, row), GPR);
      if (m0 + cg * EPP < I.lda) {                        // columns past the row end: zeros, not the next row's data
        src[i] = abase + ((int64_t)(r_begin + row) * I.lda + m0 + cg * EPP) * ES;
        inc[i] = KR * I.lda * ES; vld[i] = chunk_range(r_begin + row, r_end);
      }
    } else if (p < P) {
      // (both segments' fields are read with uniform indices and selected per lane: a divergent index into the kernel
      // argument would turn every field into a vector load whose first use -- inside the time loop -- waits vmcnt(0))
      const bool s1 = p >= pa + p0;
      const unsigned char* sp = reinterpret_cast<const unsigned char*>(s1 ? I.seg[1].p : I.seg[0].p);
      const int sld = s1 ? I.seg[1].ld : I.seg[0].ld, sn = s1 ? n1 : n0, sc0 = s1 ? I.seg[1].col0 : I.seg[0].col0;
      const int ssh = s1 ? I.seg[1].shift : I.seg[0].shift, srows = s1 ? I.seg[1].rows : I.seg[0].rows;
      const int pp = p - pa - (s1 ? p0 : 0), gpr = sn / EPP;
      // pp / gpr by a multiply: exact while pp x gpr < 2^20 (pp < 128 x 72 pieces of a chunk, gpr <= 144)
      const int row = (int)(((unsigned)pp * (s1 ? magic1 : magic0)) >> 20), cpos = pp - row * gpr;
      const int cg = F32 ? (cpos + 4 * (row & 1)) % gpr : wrap(cpos + bf_rot(sn, row), gpr);
      // row r of the chunk pairs with row r - shift of the segment
      src[i] = sp + ((int64_t)(r_begin + row - ssh) * sld + sc0 + cg * EPP) * ES;
      inc[i] = KR * sld * ES; vld[i] = chunk_range(r_begin + row - ssh, srows);
    }
    // pin the plan in registers HERE: nothing of it may still be "in flight" for the compiler when the loop starts
    asm volatile("" : "+v"(src[i]), "+v"(inc[i]), "+v"(vld[i]));
  }
  // live = false: the same NI instructions with every lane on the zero block -- the pipeline's tail keeps its instruction
  // count, so the counted vmcnt waits stay valid
  auto issue = [&](int chunk, int stage, bool live) {
    unsigned char* base = dsm + stage * stage_bytes;
#pragma unroll
    for (int i = 0; i < DWB_MAXNI; ++i) {
      if (i < NI) {                                       // NI is uniform: every wave issues the same NI instructions
        const bool ok = live && (unsigned)(chunk - (int)(vld[i] & 0xffffu)) < (vld[i] >> 16);
        const unsigned char* g = ok ? src[i] + (int64_t)chunk * inc[i] : zsrc;
        // inline asm on purpose: the compiler tracks an LDS-DMA builtin as a pending LDS write and drains it with
        // vmcnt(0) before the next LDS read -- i.e. right behind the issue, which serialises the three-stage pipeline
        // (seen in the ISA of the builtin form).  M0 = LDS byte address of THIS WAVE's lane 0 (the hardware adds 16 x the
        // lane id inside the wave, not the thread id); one wait state after the M0 write.
        const unsigned ldsaddr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void*)(base + i * DWB_THREADS * 16 + wave * 64 * 16));
        { unsigned m0_saved;    // M0 is the compiler's to manage (clobbering a reserved register is undefined behaviour): saved and restored here
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                       : "=&s"(m0_saved) : "v"(g), "s"(ldsaddr) : "memory"); }
      }
    }
  };

  // the first S - 1 chunks leave NOW: the fragment offsets and the accumulators below are set up while they are in flight
  const int S = I.stages;
  const int wait_n = (S - 2) * NI;                          // instructions that may stay outstanding when chunk c is needed
  for (int k = 0; k < S - 1; ++k) issue(k, k, k < n_chunks);

  // ---- fragment read offsets inside a stage (bytes)
  // bf16: lane (bi, q) reads row 4q + bi/4, columns c .. c+3 with c = col0 + 4 (bi%4) (transposing read; second read +16 rows)
  // F32:  lane (bi, q) reads element [row q (+ 4 per k-step)][column col0 + bi], the column group rotated like the DMA did
  const int rrow = 4 * q + (bi >> 2), rcol = 4 * (bi & 3);
  auto f32_pos = [&](int width, int col0) {           // float index of (row q, column col0 + bi) in a [16][width] slab
    const int gpr = width / 4;
    return q * width + ((col0 / 4 - 4 * (q & 1) + gpr) % gpr) * 4 + bi;
  };
  auto bf_pos = [&](int width, int col) {            // bf16 element index of (row rrow, column col) in a [32][width] slab
    const int gpr = width >> 3, rot = bf_rot(width, rrow);
    return rrow * width + wrap((col >> 3) - rot + gpr, gpr) * 8 + (col & 7);
  };
  int a_off[MF];
#pragma unroll
  for (int i = 0; i < MF; ++i)
    a_off[i] = F32 ? f32_pos(MT, wm * 16 * MF + i * 16) * 4 : bf_pos(MT, wm * 16 * MF + i * 16 + rcol) * 2;
  int b_off[NFW], b_r16[NFW];               // b_r16: bf16: bytes of 16 slab rows; F32: bytes of one k-step (4 rows)
  int njw = 0;
#pragma unroll
  for (int j = 0; j < NFW; ++j) {
    const int nf = wn + 4 * j;
    int n = nf * 16;
    if (nf < NF) njw = j + 1;
    else n = 0;                                // (read ahead unconditionally below: a fragment this wave does not own reads fragment 0)
    if (n < n0) {
      b_off[j] = pa * 16 + (F32 ? f32_pos(n0, n) * 4 : bf_pos(n0, n + rcol) * 2);
      b_r16[j] = F32 ? 4 * n0 * 4 : 16 * n0 * 2;
    } else {
      n -= n0;
      const int w1 = max(n1, 16), nc = min(n, max(n1 - 16, 0));
      b_off[j] = (pa + p0) * 16 + (F32 ? f32_pos(w1, nc) * 4 : bf_pos(w1, nc + rcol) * 2);
      b_r16[j] = F32 ? 4 * w1 * 4 : 16 * w1 * 2;
    }
  }
  njw = __builtin_amdgcn_readfirstlane(njw);

  // column sums of A (the bias gradients): the four waves that share an A fragment row (wn = 0..3) take BPW of its MF fragments
  // each -- wave wn the fragments wn BPW .. wn BPW + BPW - 1 -- instead of wave 0 all of them (8 instead of 32 accumulator
  // registers in the 256-column body, and the extra MFMAs spread over the waves)
  constexpr int BPW = (MF + 3) / 4;
  f32x4 acc[MF][NFW], accb[BPW];
#pragma unroll
  for (int b = 0; b < BPW; ++b) accb[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < MF; ++i) {
#pragma unroll
    for (int j = 0; j < NFW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
  const bool want_bias = I.cb != nullptr;                         // wave-uniform

  // ---- pipeline: S - 1 chunks in flight while chunk c is multiplied.  S (DwbItem::stages) grows as the chunk image
  // shrinks: what bounds a narrow item (a decoder: 8 KB per chunk) is the DMA latency per chunk divided by the chunks in
  // flight, not bandwidth -- with 3 stages the 22 narrow M-tiles of the MOSI plan spent ~1 us per 8-24 KB chunk.
  // (the first S - 1 chunks were requested right behind the DMA plan, ahead of the fragment offsets: see there)
  int stage = 0, nxt = S - 1;
#ifdef MFM_DWB_STAMP
  // debug build (scripts/dwb_chunk_timeline.sh): shader-clock time per phase of the chunk loop, summed over the chunks, wave 0
  long long st_wait = 0, st_bar = 0, st_issue = 0, st_math = 0;
  const long long st_begin = __builtin_readcyclecounter();
#define DWB_PT(acc) do { asm volatile("" ::: "memory"); const long long now_ = __builtin_readcyclecounter(); acc += now_ - st_last; st_last = now_; } while (0)
  long long st_last = st_begin;
#else
#define DWB_PT(acc) do { } while (0)
#endif
  for (int c = 0; c < n_chunks; ++c) {
    // chunk c is the oldest outstanding group: wait until only the S - 2 younger groups' instructions remain
    switch (wait_n) {
#define MFM_DWB_W(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
      MFM_DWB_W(1) MFM_DWB_W(2) MFM_DWB_W(3) MFM_DWB_W(4) MFM_DWB_W(5) MFM_DWB_W(6) MFM_DWB_W(7) MFM_DWB_W(8)
      MFM_DWB_W(9) MFM_DWB_W(10) MFM_DWB_W(11) MFM_DWB_W(12) MFM_DWB_W(13) MFM_DWB_W(14) MFM_DWB_W(15) MFM_DWB_W(16)
      MFM_DWB_W(17) MFM_DWB_W(18) MFM_DWB_W(19) MFM_DWB_W(20) MFM_DWB_W(21) MFM_DWB_W(22) MFM_DWB_W(23) MFM_DWB_W(24)
#undef MFM_DWB_W
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
    DWB_PT(st_wait);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // every wave's DMA of chunk c landed; chunk c-1 is consumed
    DWB_PT(st_bar);
    // The next chunk's DMA.  Issuing costs a wave ~0.2 k cycles per instruction (scripts/dwb_chunk_timeline.sh; early-fusion item,
    // per chunk: wait 0.5 k, barrier 0.4 k, issue 0.9 k, fragments + MFMA 1.4 k cycles), during which it stands still.  Measured:
    // the instructions dealt out between the MFMA groups of every wave: launch 150 -> 178 us (the stall moves into the MFMA
    // sequence); DWB_STAGGER -- waves 0-3 issue before their fragments + MFMAs, waves 4-7 after theirs, so each SIMD's two waves
    // are in different phases: 151 -> 148 us, kept.
    const int dma_chunk = c + S - 1, dma_stage = nxt;
    const bool dma_live = dma_chunk < n_chunks;
    if constexpr (F32 || !DWB_STAGGER) issue(dma_chunk, dma_stage, dma_live);
    else { if (wm == 0) issue(dma_chunk, dma_stage, dma_live); }
    DWB_PT(st_issue);
    const unsigned char* st = dsm + stage * stage_bytes;
    nxt = stage;
    stage = (stage + 1 == S) ? 0 : stage + 1;
    if constexpr (F32) {
#pragma unroll
      for (int ks = 0; ks < KC / 4; ++ks) {
        float af[MF];
#pragma unroll
        for (int i = 0; i < MF; ++i) af[i] = *reinterpret_cast<const float*>(st + a_off[i] + ks * (4 * MT * 4));
        if (want_bias) {
#pragma unroll
          for (int w = 0; w < 4; ++w)
            if (wn == w) {                                  // scalar branch
#pragma unroll
              for (int b = 0; b < BPW; ++b)
                if (w * BPW + b < MF) accb[b] = mma16x16x4(af[w * BPW + b], 1.0f, accb[b]);
            }
        }
#pragma unroll
        for (int j = 0; j < NFW; ++j) {
          if (j < njw) {                                    // scalar branch: njw is wave-uniform
            const float bf = *reinterpret_cast<const float*>(st + b_off[j] + ks * b_r16[j]);
#pragma unroll
            for (int i = 0; i < MF; ++i) acc[i][j] = mma16x16x4(af[i], bf, acc[i][j]);
          }
        }
      }
    } else {
#pragma unroll 1
      for (int kb = 0; kb < kpc; ++kb) {                    // the chunk's 32-row k-blocks
        bf16x8 af[MF];
#pragma unroll
        for (int i = 0; i < MF; ++i) af[i] = tr_frag(st + kb * (32 * MT * 2), a_off[i], 16 * MT * 2);
        // right-hand-side fragments one ahead of their MFMAs: group j's block requests group j + 1's fragment before it multiplies
        // (a wave reads at most one fragment it does not use; the reads younger than the awaited one are all in the same block, so
        // the waits stay counted across the scalar branches).  Before: read -> wait -> multiply per group, ~1.6 k cycles per
        // k-block of 32 MFMAs.
        bf16x8 bfn = tr_frag(st + kb * 2 * b_r16[0], b_off[0], b_r16[0]);
#pragma unroll
        for (int j = 0; j < NFW; ++j) {
          if (j < njw) {                                    // scalar branch: njw is wave-uniform
            const bf16x8 bf = bfn;
            if (j + 1 < NFW) bfn = tr_frag(st + kb * 2 * b_r16[j + 1], b_off[j + 1], b_r16[j + 1]);
#pragma unroll
            for (int i = 0; i < MF; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf, acc[i][j], 0, 0, 0);
          }
        }
        // (behind the products: ahead of them the block made every wave wait for all of its A fragments before the first
        // right-hand-side fragment was even requested)
        if (want_bias) {
#pragma unroll
          for (int w = 0; w < 4; ++w)
            if (wn == w) {                                  // scalar branch
#pragma unroll
              for (int b = 0; b < BPW; ++b)
                if (w * BPW + b < MF) accb[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[w * BPW + b], ones, accb[b], 0, 0, 0);
            }
        }
      }
      if constexpr (DWB_STAGGER) { if (wm != 0) issue(dma_chunk, dma_stage, dma_live); }
    }
    DWB_PT(st_math);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the trailing dummies must land before the workgroup's LDS is released
#ifdef MFM_DWB_STAMP
  if (tid == 0 && (blockIdx.x % 29) == 0)
    printf("dwb wg %3d item M=%3d N=%3d mt=%d kpc=%d chunks %3d S=%d NI=%d | prologue %6lld | per chunk: wait %5lld barrier %5lld issue %5lld math %5lld | total %7lld\n",
           (int)blockIdx.x, I.M, n0 + n1, MT, kpc, n_chunks, S, NI, st_begin - st_kernel, st_wait / n_chunks, st_bar / n_chunks,
           st_issue / n_chunks, st_math / n_chunks, (long long)__builtin_readcyclecounter() - st_kernel);
#endif

  // ---- add the tile into the gradient buffers.  Accumulator lane: rows (A columns) 4q + r, column bi of fragment j.
  if (L.debug_no_epilogue == 1) return;     // (tuning aid: MFM_DWB_NOEPI=1 measures the streaming part alone)
  if (L.slabs) {
    // slab form: the partial tile [MT][npad] of this (M-tile, row range) leaves with plain stores; column npad - 16 holds the
    // column sums of A (bias gradients).  dw_reduce_kernel adds the row ranges up.
    float* slab = L.slabs + I.slab_off + (int64_t)local * MT * I.npad;
#pragma unroll
    for (int j = 0; j < NFW; ++j) {
      if (j >= njw) continue;
      const int n = (wn + 4 * j) * 16 + bi;
#pragma unroll
      for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(int64_t)(wm * 16 * MF + i * 16 + 4 * q + r) * I.npad + n] = acc[i][j][r];
    }
    if (want_bias && bi == 0) {
#pragma unroll
      for (int b = 0; b < BPW; ++b) {
        const int i = wn * BPW + b;
        if (i >= MF) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(int64_t)(wm * 16 * MF + i * 16 + 4 * q + r) * I.npad + I.npad - 16] = accb[b][r];
      }
    }
    return;
  }
  const int Hp = I.Hp, h = I.h;
#pragma unroll
  for (int j = 0; j < NFW; ++j) {
    if (j >= njw) continue;
    const int n = (wn + 4 * j) * 16 + bi;
    float* dst = nullptr; float* dst2 = nullptr; int ldc = 0, col = 0;
#pragma unroll
    for (int o = 0; o < MFM_DWB_MAXOUT; ++o) {
      if (o < I.nout && n >= I.out[o].n0 && n < I.out[o].n0 + I.out[o].nvalid) {
        dst = I.out[o].c; dst2 = I.out[o].c2; ldc = I.out[o].ldc; col = n - I.out[o].n0;
      }
    }
    if (!dst) continue;
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 16 * MF + i * 16 + 4 * q + r;
        if (m >= I.M) continue;
        const int g = m / Hp, u = m - g * Hp;
        if (u >= h) continue;
        const int64_t o = (int64_t)(g * h + u) * ldc + col;
        atomicAdd(dst + o, acc[i][j][r]);
        if (dst2) atomicAdd(dst2 + o, acc[i][j][r]);
      }
  }
  if (want_bias && bi == 0) {
#pragma unroll
    for (int b = 0; b < BPW; ++b) {
      const int i = wn * BPW + b;
      if (i >= MF) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 16 * MF + i * 16 + 4 * q + r;
        if (m >= I.M) continue;
        const int g = m / Hp, u = m - g * Hp;
        if (u >= h) continue;
        atomicAdd(I.cb + g * h + u, accb[b][r]);
        if (I.cb2) atomicAdd(I.cb2 + g * h + u, accb[b][r]);
      }
    }
  }
}

template <bool F32, int MF, int NFW>
__global__ __launch_bounds__(DWB_THREADS) void dw_stream_kernel(const DwbLaunch L) {
  const DwbTile t = dwb_tile(L);
  dw_stream_body<F32, MF, NFW>(L, L.it[t.item], t.mt, t.sp, t.local);
}

// Two tile shapes in one launch (round 5): an entry's `mt` says which body its workgroups run.  What bounds the main loop is the
// bytes a CU takes in per row, (MT + N) x 2 (profiles/r04_bf16_large_batch.txt), and the accumulators of a workgroup hold ~74 k
// elements either way: a 256 x 256 tile moves 512 columns per row for them, a 128 x 512 one 640.  The launcher picks per item and
// splits a right-hand side of more than 256 columns into column parts.
template <int NFW128>                       // N fragments per wave of the 128-column body: 8 (N <= 512) or 9
__global__ __launch_bounds__(DWB_THREADS) void dw_stream_mixed_kernel(const DwbLaunch L_arg) {
  // The descriptor is read straight from the kernel-argument segment: with both bodies inlined the compiler no longer removed
  // its private copy of the by-value argument (3.5 KB of scratch per lane, every field a scratch load).
  typedef __attribute__((address_space(4))) const DwbLaunch kernarg_launch;
  const DwbLaunch& L = *(const DwbLaunch*)(kernarg_launch*)__builtin_amdgcn_kernarg_segment_ptr();
  const DwbTile t = dwb_tile(L);
  if (L.it[t.item].mt == 256) dw_stream_body<false, 8, 4>(L, L.it[t.item], t.mt, t.sp, t.local);
  else dw_stream_body<false, 4, NFW128>(L, L.it[t.item], t.mt, t.sp, t.local);
}

// Second launch of the slab form: block = one row m of one M-tile (an A column = a gate unit), threads = 4-column groups of the
// right-hand side; sums that row over the row ranges' slabs (coalesced 16-byte loads, the slabs are hot in L2 / MALL) and adds
// the result into the gradient tensors -- every element is owned by exactly one thread: no atomics.
__global__ __launch_bounds__(256) void dw_reduce_kernel(const DwbLaunch L) {
  // two rows per block: threads [0, 128) row 2 b, [128, 256) row 2 b + 1
  const int b = 2 * blockIdx.x + (threadIdx.x >> 7);
  if (b >= L.red_rows) return;
  int it = 0;
#pragma unroll
  for (int i = 1; i < MFM_DWB_MAXI; ++i) it += (i < L.n_items && b >= L.it[i].red_begin) ? 1 : 0;
  const DwbItem& I = L.it[it];
  const int MT = I.mt;
  const int row = b - I.red_begin;                 // A column index inside the item (padded to whole M-tiles)
  const int mt = row / MT, ml = row - mt * MT;
  const int m = row;
  if (m >= I.M) return;
  const int Hp = I.Hp, h = I.h;
  const int g = m / Hp, u = m - g * Hp;
  if (u >= h) return;
  const int npad = I.npad;
  const float* base = L.slabs + I.slab_off + ((int64_t)mt * MT + ml) * npad;       // slab of row range 0 (local = mt)
  const int64_t sstride = (int64_t)I.m_tiles * MT * npad;                           // next row range: local += m_tiles
  const int nsp = I.splits;
  for (int c4 = threadIdx.x & 127; c4 < npad / 4; c4 += 128) {
    const int n = 4 * c4;
    if (n > npad - 16) continue;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    const float* p = base + n;
    int sp = 0;
    for (; sp + 4 <= nsp; sp += 4) {                  // four independent loads in flight per thread
      const f32x4 a = *reinterpret_cast<const f32x4*>(p + (sp + 0) * sstride), b2 = *reinterpret_cast<const f32x4*>(p + (sp + 1) * sstride);
      const f32x4 c = *reinterpret_cast<const f32x4*>(p + (sp + 2) * sstride), d = *reinterpret_cast<const f32x4*>(p + (sp + 3) * sstride);
      s0 += a; s1 += b2; s2 += c; s3 += d;
    }
    for (; sp < nsp; ++sp) s0 += *reinterpret_cast<const f32x4*>(p + sp * sstride);
    const f32x4 sum = (s0 + s1) + (s2 + s3);
    if (n == npad - 16) {                           // the column sums of A
      if (I.cb) { I.cb[g * h + u] += sum[0]; if (I.cb2) I.cb2[g * h + u] += sum[0]; }
      continue;
    }
#pragma unroll
    for (int o = 0; o < MFM_DWB_MAXOUT; ++o) {
      if (o < I.nout && n >= I.out[o].n0 && n < I.out[o].n0 + I.out[o].nvalid) {
        float* dst = I.out[o].c + (int64_t)(g * h + u) * I.out[o].ldc + (n - I.out[o].n0);
        float* dst2 = I.out[o].c2 ? I.out[o].c2 + (int64_t)(g * h + u) * I.out[o].ldc + (n - I.out[o].n0) : nullptr;
        const int left = I.out[o].n0 + I.out[o].nvalid - n;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (e < left) { dst[e] += sum[e]; if (dst2) dst2[e] += sum[e]; }
      }
    }
  }
}

// x [rows, D] fp32 -> the bf16 image the kernel above streams: every modality slice padded to a multiple of 16 columns
// (pad columns zero), row stride ldo
struct XcvtArgs { const float* x; __bf16* out; int64_t rows; int D, ldo; int src0[3], n[3], dst0[3]; };
__global__ __launch_bounds__(256) void x_to_bf16_kernel(const XcvtArgs A) {
  const int64_t groups_per_row = A.ldo / 8;
  const int64_t total = A.rows * groups_per_row;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / groups_per_row;
    const int c8 = (int)(i - r * groups_per_row) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c8 + e;
      int src = -1;
#pragma unroll
      for (int s = 0; s < 3; ++s)
        if (c >= A.dst0[s] && c < A.dst0[s] + A.n[s]) src = A.src0[s] + (c - A.dst0[s]);
      v[e] = src >= 0 ? A.x[r * A.D + src] : 0.0f;
    }
    const f32x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
    const bf16x4 a = __builtin_convertvector(lo, bf16x4), b = __builtin_convertvector(hi, bf16x4);
    *reinterpret_cast<bf16x8*>(A.out + r * A.ldo + c8) = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  }
}

}  // namespace

int x_to_bf16_launch(const float* x, void* out, int64_t rows, int D, int ldo, const int* src0, const int* n, const int* dst0,
                     hipStream_t stream) {
  MFM_REQUIRE(x && out && rows >= 1 && (ldo & 7) == 0, "x_to_bf16: bad arguments");
  XcvtArgs A;
  A.x = x; A.out = reinterpret_cast<__bf16*>(out); A.rows = rows; A.D = D; A.ldo = ldo;
  for (int s = 0; s < 3; ++s) { A.src0[s] = src0[s]; A.n[s] = n[s]; A.dst0[s] = dst0[s]; }
  const int64_t total = rows * (ldo / 8);
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
  MFM_LAUNCH_TIMED(x_to_bf16_kernel, dim3(blocks), dim3(256), 0, stream, A);
  MFM_LAUNCH_CHECK("x_to_bf16_kernel");
  return MFM_OK;
}

int dw_bf16_supported(const DwbItem& I, int f32) {
  if (!I.a || I.nseg < 1 || I.nseg > 2 || I.nout < 1 || I.nout > MFM_DWB_MAXOUT) return 0;
  int np = 0;
  for (int s = 0; s < I.nseg; ++s) {
    if (!I.seg[s].p || (I.seg[s].ncols & 15) || I.seg[s].ncols < 16) return 0;
    // bf16 slabs: 16-byte pieces of 8 elements on 16-byte boundaries; fp32: any dword-aligned column range of the batch
    // (global_load_dwordx4 ... lds takes dword-aligned sources: tests with ld = 325, first column 305)
    if (!f32 && ((I.seg[s].ld & 7) || (I.seg[s].col0 & 7) || (((uintptr_t)I.seg[s].p) & 15) != 0)) return 0;
    if (f32 && (((uintptr_t)I.seg[s].p) & 3) != 0) return 0;
    np += I.seg[s].ncols;
  }
  if (np > 16 * 4 * DWB_NFW) return 0;        // (the general form; the launcher decides about the 128-column one)
  if ((I.lda & (f32 ? 3 : 7)) || (((uintptr_t)I.a) & 15) != 0 || I.M < 1 || I.Hp < I.h || I.h < 1) return 0;
  const int P = DWB_KC * (DWB_MT + np) / 8;
  if ((P + DWB_THREADS - 1) / DWB_THREADS > DWB_MAXNI) return 0;
  return 1;
}

// 256 bytes of zeros per device for the DMA lanes that must read as zero (allocated once per device, never freed: the one
// exception to "the caller owns every buffer" in this library, next to the P2P staging blocks)
static const void* zero_block() {
  static void* z[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!z[dev]) {
    void* p = nullptr;
    if (hipMalloc(&p, 256) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 256) != hipSuccess) { (void)hipFree(p); return nullptr; }
    z[dev] = p;
  }
  return z[dev];
}

static int item_cols(const DwbItem& I) {
  int N = 0;
  for (int s = 0; s < I.nseg; ++s) N += I.seg[s].ncols;
  return N;
}
// What a workgroup pays per 32-row chunk, relative.  128- / 256-column tile forms (round 5, scripts/dwb_chunk_timeline.sh, shader
// clocks per chunk at MOSI sizes, B = 2048): N = 32 / 112 (2 DMA instructions per thread) 1.81 k / 1.84 k, N = 336 (4) 2.75 k,
// N = 480 (5) 3.28 k; 256-column tiles: N = 112 (3) 2.41 k, N = 240 (4) 2.97 k -- i.e. ~0.9 k + 0.48 k per DMA instruction (8 KB
// into the CU at ~16 B/clk), whatever the MFMA count.  The older model (1 + N / 128, kept for the 96-column form) gave narrow
// items too few workgroups: theirs ran 250 k cycles while the early-fusion item's finished after 180 k.
// kpc: 32-row k-blocks per chunk; the cost is per 32 rows.  Infinite when the chunk needs more than DWB_MAXNI instructions.
static double chunk_cost(int mt, int N, bool wide, int kpc = 1) {
  if (!wide) return 1.0 + N / 128.0;
  const int NI = (DWB_KC * kpc * (mt + N) / 8 + DWB_THREADS - 1) / DWB_THREADS;
  if (NI > DWB_MAXNI) return 1e30;
  return (1.9 + NI) / kpc;
}
// rows per chunk of an item (128-column forms): what costs least per row, as long as a row range keeps >= 8 chunks
static int best_kpc(int mt, int N, bool wide, int rows) {
  int best = 1;
  const char* e = opt_get("MFM_DWB_KPC");
  const int cap = e ? atoi(e) : 4;
  for (int k = 2; k <= cap && wide; k *= 2)
    if (chunk_cost(mt, N, wide, k) < chunk_cost(mt, N, wide, best) && rows >= 8 * DWB_KC * k) best = k;
  return best;
}

int dw_bf16_launch(DwbLaunch& Lu, hipStream_t stream) {
  MFM_REQUIRE(Lu.n_items >= 1 && Lu.n_items <= MFM_DWB_MAXI && Lu.rows >= 1, "dw bf16: bad launch");
  if (!Lu.zeros) Lu.zeros = zero_block();
  Lu.debug_no_epilogue = opt_get("MFM_DWB_NOEPI") ? atoi(opt_get("MFM_DWB_NOEPI")) : 0;
  MFM_REQUIRE(Lu.zeros, "dw bf16: no zero block");
  DwbLaunch L = Lu;                       // working copy: entries may be split into column parts below
  for (int i = 0; i < L.n_items; ++i) { L.it[i].parts = 1; L.it[i].part = 0; }
  double wsum = 0.0;
  for (int i = 0; i < L.n_items; ++i) {
    DwbItem& I = L.it[i];
    MFM_REQUIRE(dw_bf16_supported(I, L.f32), "dw one-pass: item %d is not supported", i);
    const int64_t es = L.f32 ? 4 : 2;
    MFM_REQUIRE((int64_t)4 * DWB_KC * I.lda * es < ((int64_t)1 << 31), "dw one-pass: A row stride too large");
    for (int s = 0; s < I.nseg; ++s)
      MFM_REQUIRE((int64_t)4 * DWB_KC * I.seg[s].ld * es < ((int64_t)1 << 31) && I.seg[s].shift >= 0, "dw one-pass: segment %d row stride too large", s);
  }
  // tile shape: 128-column M-tiles with <= 512 right-hand-side columns when every item fits (bf16 form only), else 96 / 576
  // (measured, MOSI sizes: T*B = 40960 rows 180 vs 173 us -- the streaming part is 24 us shorter, 134 vs 158, but the same
  // number of workgroups now adds 128-column partial tiles: 46 instead of 15 us of atomics -- T*B = 81920 rows 269 vs 309 us:
  // the wide form from 65536 rows on; MFM_DWB_MF=4 / 3 forces one)
  const int mf_env = opt_get("MFM_DWB_MF") ? atoi(opt_get("MFM_DWB_MF")) : 0;
  // (slab form: the 128-column tiles' larger partial tiles cost plain stores, not atomics -> wide from 8192 rows on)
  // measured (profiles/r04_bf16_large_batch.txt): ahead up to T*B = 40960 rows, behind from 81920 (MFM_DWB_SLABS=1 / 0 forces)
  const char* slab_env = opt_get("MFM_DWB_SLABS");
  const bool slab_req = L.slabs != nullptr && !L.f32 && (slab_env ? atoi(slab_env) != 0 : L.rows <= 65536);
  // (round 5, with the per-item shapes / 64-128-row chunks / cost model below: ahead from the smallest bf16-resident plan on,
  // bench.py --dtype bf16, ms per step, 128-column form forced vs 96-column form: B = 128 0.2717 vs 0.2752, 192 0.2827 vs 0.2853,
  // 256 0.2925 vs 0.2975, 384 0.3151 vs 0.3276 -> from 2048 rows)
  bool wide = !L.f32 && mf_env != 3 && (mf_env == 4 || L.rows >= (slab_req ? 2048 : 65536));
  bool wide9 = false;
  for (int i = 0; i < L.n_items && wide; ++i) {
    int N = 0;
    for (int s = 0; s < L.it[i].nseg; ++s) N += L.it[i].seg[s].ncols;
    wide = N <= 16 * 4 * 9 && (DWB_KC * (128 + N) / 8 + DWB_THREADS - 1) / DWB_THREADS <= DWB_MAXNI;
    if (N > 16 * 4 * 8) wide9 = true;              // right-hand sides of 513-576 columns: the (4, 9) instantiation
  }
  for (int i = 0; i < L.n_items; ++i) L.it[i].mt = wide ? 128 : DWB_MT;
  // Tile shape per item (round 5, 128-column form only; MFM_DWB_MIXED=0 keeps one shape): the main loop is bound by the bytes a
  // CU takes in, (MT + N) x 2 per row and tile -- a 256 x <= 256 tile (8 x 4 fragments per wave, the same 32-36 accumulator
  // tiles) moves fewer columns than 128 x N when the item is tall and wide: the early-fusion encoder, M = 512 N = 480: 4 x 608
  // -> 2 M-tiles x 2 column parts x 496; a decoder, M = 448 N = 112: 4 x 240 -> 2 x 368.  Taken when it saves >= 5 %.
  bool mixed = false;
  {
    const char* me = opt_get("MFM_DWB_MIXED");
    if (wide && !(me && atoi(me) == 0)) {
      DwbLaunch X = L;
      X.n_items = 0;
      for (int i = 0; i < L.n_items; ++i) {
        const DwbItem& I = L.it[i];
        const int n0 = I.seg[0].ncols, n1 = I.nseg > 1 ? I.seg[1].ncols : 0, N = n0 + n1;
        const int k = (N + 255) / 256, nf = N / 16;
        const double c128 = (I.M + 127) / 128 * chunk_cost(128, N, true, best_kpc(128, N, true, L.rows));
        double c256 = 0.0;
        for (int p = 0; p < k; ++p) {
          const int np = 16 * (nf / k + (p < nf % k ? 1 : 0));
          c256 += (I.M + 255) / 256 * chunk_cost(256, np, true, best_kpc(256, np, true, L.rows));
        }
        const int left = L.n_items - i - 1;
        const bool take = I.M > 128 && c256 <= c128 * 0.95 && X.n_items + k + left <= MFM_DWB_MAXI;
        if (!take) { X.it[X.n_items++] = I; continue; }
        mixed = true;
        int cs = 0;
        for (int p = 0; p < k; ++p) {
          const int ce = cs + 16 * (nf / k + (p < nf % k ? 1 : 0));
          DwbItem& J = X.it[X.n_items++];
          J = I;
          J.mt = 256; J.parts = k; J.part = p;
          if (p > 0) { J.cb = nullptr; J.cb2 = nullptr; }          // the column sums of A: the first part's
          J.nseg = 0; J.nout = 0;
          if (cs < n0) { DwbSeg& S = J.seg[J.nseg++]; S = I.seg[0]; S.col0 += cs; S.ncols = std::min(ce, n0) - cs; }
          if (ce > n0) { DwbSeg& S = J.seg[J.nseg++]; S = I.seg[1]; S.col0 += std::max(cs, n0) - n0; S.ncols = ce - std::max(cs, n0); }
          for (int o = 0; o < I.nout; ++o) {
            const int lo = std::max(I.out[o].n0, cs), hi = std::min(I.out[o].n0 + I.out[o].nvalid, ce);
            if (lo >= hi) continue;
            DwbOut& O = J.out[J.nout++];
            O = I.out[o];
            O.n0 = lo - cs; O.nvalid = hi - lo; O.c = I.out[o].c + (lo - I.out[o].n0);
            O.c2 = I.out[o].c2 ? I.out[o].c2 + (lo - I.out[o].n0) : nullptr;
          }
          cs = ce;
        }
      }
      if (mixed) L = X;
    }
  }
  for (int i = 0; i < L.n_items; i += L.it[i].parts) {       // rows per chunk: one value per item (the parts share the row ranges)
    int kpc = 4;
    for (int p = 0; p < L.it[i].parts; ++p) kpc = std::min(kpc, L.f32 ? 1 : best_kpc(L.it[i + p].mt, item_cols(L.it[i + p]), wide, L.rows));
    for (int p = 0; p < L.it[i].parts; ++p) L.it[i + p].kpc = kpc;
  }
  auto item_cost = [&](const DwbItem& I) { return chunk_cost(I.mt, item_cols(I), wide, I.kpc); };
  for (int i = 0; i < L.n_items; ++i) {
    DwbItem& I = L.it[i];
    I.m_tiles = (I.M + I.mt - 1) / I.mt;
    wsum += (double)I.m_tiles * item_cost(I);
  }
  // one workgroup per CU (the chunk stages fill the LDS), so the launch runs in whole ROUNDS of workgroups: row ranges sized
  // so that it fills `rounds` rounds and not one workgroup more -- a range's cost taken as (fixed part + N / 128) per chunk
  // (profiles/r02_dw_onepass.txt).  One round up to 32768 rows, two above: every workgroup ends with its partial tile's
  // atomics, so few long ranges win while the rows are few (measured, MOSI sizes, T*B = 5120 / 10240 / 20480 / 40960 rows,
  // us at 0.5 / 0.75 / 1 / 2 x CUs workgroups: 56 / 59 / 89 / 99, 94 / 73 / 106 / 107, 175 / 122 / 145 / 125, 335 / 229 / 267 / 173:
  // "1 x CUs" came out at a few workgroups more than CUs and ran two rounds).  MFM_DWB_TARGET = workgroups / CUs (no fitting).
  const int cus = device_cus();
  // Round 5 (after the recurrences got shorter the launch was measured again, MFM_DWB_TARGET sweep, ms per step, 1 / 2 / 3 rounds):
  // 128-column tiles (MOSI sizes):  T*B = 40960: 0.5646 / 0.5870 / 0.5833,  81920: 0.9817 / 1.0073 / 1.0238 -> ONE round
  // (half the partial tiles: 35 instead of 71 MB of slabs written and read back).  The MOSEI / YouTube sizes (right-hand sides of
  // 544 columns) then got a (4, 9) instantiation of the 128-column form -- 232 VGPRs, no spills -- instead of 96-column tiles in
  // two rounds: MOSEI T = 50 B = 1024 0.833 -> 0.811 ms, YouTube T = 50 B = 2048 1.300 -> 1.258 ms, MOSEI T = 20 B = 2048
  // 0.656 -> 0.631 ms.  96-column tiles (right-hand sides beyond 576 columns) keep two rounds above 32768 rows.
  const int rounds = (L.rows <= 32768 || wide) ? 1 : 2;                 // (81920 rows run the atomics form: measured with it)
  const char* tenv = opt_get("MFM_DWB_TARGET");
  int tiles = 0;
  size_t smem = 0;
  for (double fill = 0.98; ; fill -= 0.04) {
    const double target = tenv ? atof(tenv) * cus : fill * rounds * cus;
    tiles = 0;
    for (int i = 0; i < L.n_items; i += L.it[i].parts) {         // an item = `parts` consecutive entries with one set of row ranges
      const int k = L.it[i].parts;
      double c = 0.0;
      for (int p = 0; p < k; ++p) c += item_cost(L.it[i + p]) / k;
      int splits = (int)(target * c / wsum + 0.5);
      const int KR = DWB_KC * L.it[i].kpc;
      const int max_splits = std::max(1, L.rows / (4 * KR));
      splits = std::max(1, std::min(splits, max_splits));
      const int rps = ((L.rows + splits - 1) / splits + KR - 1) / KR * KR;
      for (int p = 0; p < k; ++p) {
        DwbItem& I = L.it[i + p];
        I.rows_per_split = rps;
        I.splits = (L.rows + rps - 1) / rps;
        I.tile_begin = p == 0 ? tiles : INT_MAX;
      }
      tiles += k * L.it[i].m_tiles * L.it[i].splits;
    }
    if (tenv || tiles <= rounds * cus || fill < 0.3) break;
  }
  for (int i = 0; i < L.n_items; ++i) {
    DwbItem& I = L.it[i];
    int N = 0;
    for (int s = 0; s < I.nseg; ++s) N += I.seg[s].ncols;
    const int P = DWB_KC * I.kpc * (I.mt + N) / 8;
    const int NI = (P + DWB_THREADS - 1) / DWB_THREADS;
    MFM_REQUIRE(NI <= DWB_MAXNI, "dw one-pass: %d DMA instructions per chunk", NI);
    // stages: as many as fit ~144 KB, the counted wait and the cap (MFM_DWB_STAGES forces a count, clamped)
    int S = (int)((144 * 1024) / ((size_t)NI * DWB_THREADS * 16));
    S = std::min(S, DWB_MAX_WAIT / NI + 2);
    if (const char* e = opt_get("MFM_DWB_STAGES")) S = std::min(S, atoi(e));
    S = std::max(DWB_MIN_STAGES, std::min(S, DWB_MAX_STAGES));
    I.stages = S;
    smem = std::max(smem, (size_t)S * NI * DWB_THREADS * 16);
  }
  // slab form: partial tiles into the caller's scratch, summed by a second launch (bf16 form; MFM_DWB_SLABS=0 keeps the atomics)
  bool slabs = slab_req;
  if (slabs) {
    int64_t off = 0;
    int red = 0;
    for (int i = 0; i < L.n_items; ++i) {
      DwbItem& I = L.it[i];
      int N = 0;
      for (int s = 0; s < I.nseg; ++s) N += I.seg[s].ncols;
      I.npad = N + 16;                                  // (N is a multiple of 16: the bias column starts a 64-byte group)
      I.slab_off = off;
      off += (int64_t)I.m_tiles * I.splits * I.mt * I.npad;
      I.red_begin = red;
      red += I.m_tiles * I.mt;
    }
    if (off > L.slab_floats) slabs = false;           // (scratch sized for fewer workgroups than a forced MFM_DWB_TARGET asks for)
    else L.red_rows = red;
  }
  if (!slabs) L.slabs = nullptr;
  static bool attr = false;
  if (!attr) {
    MFM_HIP_CHECK(hipFuncSetAttribute((const void*)dw_stream_kernel<false, 3, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MFM_HIP_CHECK(hipFuncSetAttribute((const void*)dw_stream_kernel<false, 4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MFM_HIP_CHECK(hipFuncSetAttribute((const void*)dw_stream_kernel<false, 4, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MFM_HIP_CHECK(hipFuncSetAttribute((const void*)dw_stream_kernel<true, 3, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MFM_HIP_CHECK(hipFuncSetAttribute((const void*)dw_stream_mixed_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MFM_HIP_CHECK(hipFuncSetAttribute((const void*)dw_stream_mixed_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  MFM_REQUIRE(smem <= 160 * 1024, "dw one-pass: %zu bytes of LDS", smem);
  // STATUS of the fp32 form: parity-green (tests/test_gpu_large_batch.py), but at B = 2048 it runs 666 us (+ 94 us of tail
  // GEMM) against 626 us for the grouped GEMM: its main loop keeps the fp32 matrix pipe ~43 % busy (one ds_read_b32 and its
  // wait per three MFMAs; batching a k-step's reads ahead of its MFMAs made it 779 us) -> opt-in, MFM_DW_F32_MINROWS
  if (L.f32) MFM_LAUNCH_TIMED((dw_stream_kernel<true, 3, 9>), dim3(tiles), dim3(DWB_THREADS), smem, stream, L);
  else if (mixed && wide9) MFM_LAUNCH_TIMED(dw_stream_mixed_kernel<9>, dim3(tiles), dim3(DWB_THREADS), smem, stream, L);
  else if (mixed) MFM_LAUNCH_TIMED(dw_stream_mixed_kernel<8>, dim3(tiles), dim3(DWB_THREADS), smem, stream, L);
  else if (wide && wide9) MFM_LAUNCH_TIMED((dw_stream_kernel<false, 4, 9>), dim3(tiles), dim3(DWB_THREADS), smem, stream, L);
  else if (wide) MFM_LAUNCH_TIMED((dw_stream_kernel<false, 4, 8>), dim3(tiles), dim3(DWB_THREADS), smem, stream, L);
  else MFM_LAUNCH_TIMED((dw_stream_kernel<false, 3, 9>), dim3(tiles), dim3(DWB_THREADS), smem, stream, L);
  MFM_LAUNCH_CHECK("dw_stream_kernel");
  if (L.slabs && L.debug_no_epilogue != 1) {
    MFM_LAUNCH_TIMED(dw_reduce_kernel, dim3((L.red_rows + 1) / 2), dim3(256), 0, stream, L);
    MFM_LAUNCH_CHECK("dw_reduce_kernel");
  }
  return MFM_OK;
}

int64_t dw_bf16_scratch_floats(int64_t rows) {
  // one workgroup per CU and round, every partial tile at most 128 x (576 + 16) floats
  const int64_t rounds = rows <= 32768 ? 1 : 2;
  return rounds * device_cus() * (int64_t)128 * 592;
}

}  // namespace mfm

// Test / tuning entry points (not used by the plan, which builds DwbLaunch itself): one LSTM's weight gradients from
// bf16-resident (f32 = 0) or fp32 (f32 = 1) buffers.  dA [rows, 4 Hp], xb [rows, ldx] (columns [xcol0, xcol0 + dx) used), hs
// [rows, Hp] (row r pairs with hs[r - shift]); outputs fp32, accumulated into.
static int dw_lstm_entry(int f32, const void* dA, int32_t rows, int32_t h, const void* xb, int32_t ldx, int32_t xcol0, int32_t dx,
                         const void* hs, int32_t shift, float* dw_ih, float* dw_hh, float* dw_hh2, float* db_ih, float* db_hh,
                         void* stream) {
  using namespace mfm;
  MFM_REQUIRE(dA && hs && dw_hh && rows >= 1 && h >= 1, "mfm_dw_*_lstm: bad arguments");
  const int Hp = round_up(h, 16);
  DwbLaunch L;
  memset(&L, 0, sizeof(L));
  L.rows = rows; L.n_items = 1; L.f32 = f32;
  DwbItem& I = L.it[0];
  I.a = reinterpret_cast<const __bf16*>(dA); I.lda = 4 * Hp; I.M = 4 * Hp; I.Hp = Hp; I.h = h;
  int n = 0;
  if (xb) {
    MFM_REQUIRE(dw_ih && dx >= 1, "mfm_dw_*_lstm: x given without dw_ih / dx");
    I.seg[I.nseg].p = reinterpret_cast<const __bf16*>(xb); I.seg[I.nseg].ld = ldx; I.seg[I.nseg].ncols = round_up(dx, 16);
    I.seg[I.nseg].col0 = xcol0; I.seg[I.nseg].shift = 0;
    // (fp32: a slab wider than what is left of the LAST row would leave the buffer: the caller's plan moves that row to a
    // K = 1 GEMM; the test entry point simply requires the slack to exist or the slab to fit)
    I.seg[I.nseg].rows = rows; ++I.nseg;
    I.out[I.nout].n0 = 0; I.out[I.nout].nvalid = dx; I.out[I.nout].c = dw_ih; I.out[I.nout].ldc = dx; ++I.nout;
    n = round_up(dx, 16);
  }
  I.seg[I.nseg].p = reinterpret_cast<const __bf16*>(hs); I.seg[I.nseg].ld = Hp; I.seg[I.nseg].ncols = Hp;
  I.seg[I.nseg].col0 = 0; I.seg[I.nseg].shift = shift; I.seg[I.nseg].rows = rows; ++I.nseg;
  I.out[I.nout].n0 = n; I.out[I.nout].nvalid = h; I.out[I.nout].c = dw_hh; I.out[I.nout].c2 = dw_hh2; I.out[I.nout].ldc = h; ++I.nout;
  I.cb = db_ih; I.cb2 = db_hh;
  if (!dw_bf16_supported(I, f32)) { set_error("mfm_dw_*_lstm: shape not supported (h=%d dx=%d ldx=%d)", h, dx, ldx); return MFM_ERR_UNSUPPORTED; }
  return dw_bf16_launch(L, (hipStream_t)stream);
}

extern "C" int mfm_dw_bf16_lstm(const void* dA, int32_t rows, int32_t h, const void* xb, int32_t ldx, int32_t dx,
                                const void* hs, int32_t shift, float* dw_ih, float* dw_hh, float* dw_hh2, float* db_ih,
                                float* db_hh, void* stream) {
  return dw_lstm_entry(0, dA, rows, h, xb, ldx, 0, dx, hs, shift, dw_ih, dw_hh, dw_hh2, db_ih, db_hh, stream);
}

extern "C" int mfm_dw_f32_lstm(const float* dA, int32_t rows, int32_t h, const float* x, int32_t ldx, int32_t xcol0, int32_t dx,
                               const float* hs, int32_t shift, float* dw_ih, float* dw_hh, float* dw_hh2, float* db_ih,
                               float* db_hh, void* stream) {
  return dw_lstm_entry(1, dA, rows, h, x, ldx, xcol0, dx, hs, shift, dw_ih, dw_hh, dw_hh2, db_ih, db_hh, stream);
}
