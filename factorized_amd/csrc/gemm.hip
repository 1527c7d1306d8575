// Grouped, batched, strided fp32 GEMM on v_mfma_f32_16x16x4_f32 (gfx950).
//
// One launch covers up to MFM_GEMM_MAXP independent problems (input projections of the 4
// encoders; all 49 weight-gradient products of the backward; ...), because at the MOSI batch
// size every one of them is far too small to fill 256 CUs on its own and a launch boundary costs
// several us (a grouped launch of this kernel has a ~7 us floor, profiles/r01 GEMM microbenchmark).  Tiles are 32x32 or 64x64 (picked so the group yields >= ~2 blocks
// per CU), K is staged 32 deep through LDS.  The LDS image follows the operand's memory order so that
// a thread's 4-element group is ONE ds_write_b128: [k][m+16] for m-/n-contiguous operands (MFMA
// operand reads conflict-free: row stride == 16 mod 32 banks), [m][k+4] for k-contiguous ones (reads
// at most 2-way).  Writing k-contiguous groups into a [k][m] image cost 8-way-conflicted ds_write_b32.
#include <stdlib.h>

#include "internal.h"
#include "gemm_common.h"

namespace mfm {

constexpr int BK = 32;   // K depth of one LDS stage: 8 MFMA k-steps between barriers

// VEC: every operand of every problem in the group is unit-stride along its 4-element load groups
// (true for all products of the MFM step); the 16-byte path is then unconditional.  A launch with
// an oddly strided operand uses the VEC=false instantiation (dword buffer loads) for the whole group.
// Occupancy target (second launch-bounds argument = waves per SIMD): a K step is a chain of latencies (LDS
// write -> barrier -> fragment reads -> 8 MFMAs), so resident waves are what keeps the MFMA pipe fed.  88
// registers / 5 workgroups per CU for the 32x32 tiles, 158 / 3 for the 64x64 tiles, no spills; measured
// against the compiler's default (108 / 4 and 188 / 2): +4 % on the whole step at B=2048, +1 % at B=32;
// 6 waves (80 registers, 4 spilled) gains nothing more.
// FR = 4: 128x128 tiles (each wave 64x64 = 16 accumulators) for launches with thousands of 64x64 tiles: twice the MFMA
// work per staged byte and per barrier of the FR = 2 variant; 2 workgroups per CU (73 KB of LDS, <= 256 registers).
// It carries no squared-error epilogue (64 more live registers): launches that need one stay on FR <= 2.
template <int FR, bool VEC>
__global__ __launch_bounds__(256, (FR == 1) ? (VEC ? 5 : 4) : (FR == 2 ? 3 : 2)) void gemm_f32_kernel(const GemmGroup g) {
  constexpr int BM = 32 * FR, BN = 32 * FR;
  constexpr int LDA = BM + 16, LDB = BN + 16;     // [k][m] layout (m-/n-contiguous operands)
  constexpr int LDK = BK + 4;                     // [m][k] layout (k-contiguous operands)
  constexpr int EPT_A = BM * BK / 256, EPT_B = BN * BK / 256;
  constexpr int ASZ = (BK * LDA > BM * LDK) ? BK * LDA : BM * LDK;
  constexpr int BSZ = (BK * LDB > BN * LDK) ? BK * LDB : BN * LDK;
  // two LDS images per operand: while the MFMAs of K-tile kt read image kt&1, the registers of tile kt+1 are
  // written to the other one, so a K step costs ONE barrier instead of two
  __shared__ __attribute__((aligned(16))) float As2[2][ASZ];
  __shared__ __attribute__((aligned(16))) float Bs2[2][BSZ];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int bi = lane & 15, q = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- locate problem / tile (wave-uniform)
  // Workgroups are dealt to the 8 XCDs round-robin by id, and each XCD has its own L2.  Tiles that share
  // an operand panel (same m-tile across n, same n-tile across m / gates) are neighbours in the LOGICAL
  // order, so inside every problem each XCD is given one contiguous run of logical tiles: a panel is then
  // fetched from HBM once per XCD that needs it instead of once per tile (the LSTM weight-gradient launch
  // read 37.7 MB for ~6 MB of operands before this), while every XCD still gets an equal
  // share of every problem (a whole-launch remap left the XCDs with the long-K problems as stragglers).
  int pi = 0;
  const int bid = blockIdx.x;
#pragma unroll
  for (int i = 1; i < MFM_GEMM_MAXP; ++i) pi += (bid >= g.begins[i]) ? 1 : 0;
  const GemmProblem& P = g.p[pi];
  const MfmGemmDesc& d = P.d;
  int local = bid - P.block_begin;
  {
    constexpr int NX = 8;
    const int nb = ((pi + 1 < g.count) ? g.begins[pi + 1] : (int)gridDim.x) - P.block_begin;
    const int x = local % NX, j = local / NX;
    const int per = nb / NX, rem = nb % NX;
    local = x * per + (x < rem ? x : rem) + j;
  }
  const int tn = local % P.tiles_n; local /= P.tiles_n;
  const int tm = local % P.tiles_m; local /= P.tiles_m;
  const int z = local % d.batch;
  const int split = local / d.batch;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = split * P.k_per_split;
  const int kend = min(d.k, kbeg + P.k_per_split);
  if (kbeg >= kend && split > 0) return;

  const float* __restrict__ A = d.a + (int64_t)z * d.a_sz;
  const float* __restrict__ Bm = d.b + (int64_t)z * d.b_sz;
  const bool a_mcontig = (d.a_sm == 1 && d.a_sk != 1);
  const bool b_ncontig = (d.b_sn == 1 && d.b_sk != 1);

  // Register ring of DEPTH K-tiles in flight.  At these sizes (K = 300..640, 8 MFMAs per wave and
  // tile) one workgroup's K loop is a chain of latencies and, above all, of vector-memory instructions:
  // measured ~10 B/clk/CU with dword loads whatever else was tuned.  So:
  //   (a) tiles are fetched with BUFFER loads, 16 bytes per lane wherever the operand is unit-stride
  //       along the thread's 4-element group (buffer_load_dwordx4 needs only 4-byte alignment); the
  //       descriptor's num_records makes reads past the end of the operand return 0, so a group may
  //       straddle a row end or the K tail without any branch -- invalid elements are zeroed by a
  //       MULTIPLY (a select/branch makes hipcc wait vmcnt(0) right behind the load);
  //   (b) the barriers are LDS-only (lds_barrier): __syncthreads() drains the ring with vmcnt(0);
  //   (c) the operand fragments of a tile are read from LDS before the first MFMA.
  constexpr int DEPTH = (FR == 1) ? 3 : (FR == 2 ? 2 : 1);
  constexpr int GA = EPT_A / 4, GB = EPT_B / 4;
  float ra[DEPTH][EPT_A], rb[DEPTH][EPT_B];
  const int a_sm = (int)d.a_sm, a_sk = (int)d.a_sk, b_sk = (int)d.b_sk, b_sn = (int)d.b_sn;
  // extents in bytes for the bounds-checked descriptors (host guarantees < 2^31)
  const int a_bytes = ((d.m - 1) * a_sm + (max(d.k, 1) - 1) * a_sk + 1) * 4;
  const int b_bytes = ((max(d.k, 1) - 1) * b_sk + (max(d.n_valid, 1) - 1) * b_sn + 1) * 4;
  const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t bres = __builtin_amdgcn_make_buffer_rsrc((void*)Bm, 0, b_bytes, 0x00020000);
  // loop-invariant group coordinates
  int a_base[GA], a_gk[GA], a_gm[GA], b_base[GB], b_gk[GB], b_gn[GB];
#pragma unroll
  for (int g = 0; g < GA; ++g) {
    const int idx = (tid * GA + g) * 4;
    const int k = a_mcontig ? idx / BM : idx % BK;
    const int m = a_mcontig ? idx % BM : idx / BK;
    a_gk[g] = k; a_gm[g] = m0 + m;
    a_base[g] = min(m0 + m, d.m - 1) * a_sm;
  }
#pragma unroll
  for (int g = 0; g < GB; ++g) {
    const int idx = (tid * GB + g) * 4;
    const int k = b_ncontig ? idx / BN : idx % BK;
    const int n = b_ncontig ? idx % BN : idx / BK;
    b_gk[g] = k; b_gn[g] = n0 + n;
    b_base[g] = min(n0 + n, max(d.n_valid - 1, 0)) * b_sn;
  }
  const int klast = max(kend - 1, 0);
  auto load_tiles = [&](int slot, int k0) {
#pragma unroll
    for (int g = 0; g < GA; ++g) {
      const int gk = k0 + a_gk[g], gm = a_gm[g];
      f32x4 v;
      if constexpr (VEC) {
        const int off = a_base[g] + min(gk, klast) * a_sk;
        v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ares, off * 4, 0, 0));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                     ares, (a_base[g] + min(gk + e, klast) * a_sk) * 4, 0, 0));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int em = a_mcontig ? gm + e : gm, ek = a_mcontig ? gk : gk + e;
        ra[slot][4 * g + e] = v[e] * (float)((int)(em < d.m) & (int)(ek < kend));
      }
    }
#pragma unroll
    for (int g = 0; g < GB; ++g) {
      const int gk = k0 + b_gk[g], gn = b_gn[g];
      f32x4 v;
      if constexpr (VEC) {
        const int off = b_base[g] + min(gk, klast) * b_sk;
        v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bres, off * 4, 0, 0));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                     bres, (b_base[g] + min(gk + e, klast) * b_sk) * 4, 0, 0));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int en = b_ncontig ? gn + e : gn, ek = b_ncontig ? gk : gk + e;
        rb[slot][4 * g + e] = v[e] * (float)((int)(en < d.n_valid) & (int)(ek < kend));
      }
    }
  };
  // LDS strides of the two images (block-uniform values, no divergent control flow)
  const int a_lm = a_mcontig ? 1 : LDK, a_lk = a_mcontig ? LDA : 1;
  const int b_ln = b_ncontig ? 1 : LDK, b_lk = b_ncontig ? LDB : 1;
  int a_st[GA], b_st[GB];
#pragma unroll
  for (int g = 0; g < GA; ++g) {
    const int idx = (tid * GA + g) * 4;
    const int k = a_mcontig ? idx / BM : idx % BK;
    const int m = a_mcontig ? idx % BM : idx / BK;
    a_st[g] = m * a_lm + k * a_lk;
  }
#pragma unroll
  for (int g = 0; g < GB; ++g) {
    const int idx = (tid * GB + g) * 4;
    const int k = b_ncontig ? idx / BN : idx % BK;
    const int n = b_ncontig ? idx % BN : idx / BK;
    b_st[g] = n * b_ln + k * b_lk;
  }
  auto store_tiles = [&](int slot, int img) {
    float* As = As2[img];
    float* Bs = Bs2[img];
#pragma unroll
    for (int g = 0; g < GA; ++g)
      *reinterpret_cast<f32x4*>(As + a_st[g]) = f32x4{ra[slot][4 * g], ra[slot][4 * g + 1], ra[slot][4 * g + 2], ra[slot][4 * g + 3]};
#pragma unroll
    for (int g = 0; g < GB; ++g)
      *reinterpret_cast<f32x4*>(Bs + b_st[g]) = f32x4{rb[slot][4 * g], rb[slot][4 * g + 1], rb[slot][4 * g + 2], rb[slot][4 * g + 3]};
  };

  // squared-error epilogue: the targets of this thread's outputs are requested before the K loop
  const bool do_mse = (FR <= 2) && pi < g.mse_count;          // wave-uniform
  const MseEpi& me = g.mse[do_mse ? pi : 0];
  constexpr int TF = (FR <= 2) ? FR : 1;            // FR = 4 has no squared-error epilogue: no target registers
  float tgt[TF][TF][4];
  if constexpr (FR <= 2) if (do_mse) {
#pragma unroll
    for (int fm = 0; fm < FR; ++fm)
#pragma unroll
      for (int fn = 0; fn < FR; ++fn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = min(m0 + wm * 16 * FR + fm * 16 + q * 4 + r, d.m - 1);
          const int col = min(n0 + wn * 16 * FR + fn * 16 + bi, max(d.n_valid - 1, 0));
          tgt[fm][fn][r] = me.x[(int64_t)row * me.ldx + col];
        }
  }
  f32x4 acc[FR][FR];
#pragma unroll
  for (int i = 0; i < FR; ++i)
#pragma unroll
    for (int j = 0; j < FR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt = (kend - kbeg + BK - 1) / BK;
  // the ring is always filled DEPTH deep (tiles past the end load clamped addresses and are masked
  // to zero): no branch around a load anywhere
#pragma unroll
  for (int s = 0; s < DEPTH; ++s) load_tiles(s, kbeg + s * BK);
  store_tiles(0, 0);
  load_tiles(0, kbeg + DEPTH * BK);
  lds_barrier();
  // ring slot of K-tile kt is kt % DEPTH; slot 0 was just refilled with tile DEPTH
  for (int kt0 = 0; kt0 < nkt; kt0 += DEPTH) {
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
      const int kt = kt0 + s;
      const int img = kt & 1;
      const float* As = As2[img];
      const float* Bs = Bs2[img];
      // operand fragments are read from LDS ahead of the MFMAs that use them: the whole tile for the 32x32
      // variant, half a tile at a time for the 64x64 one (its 4 accumulators already fill the register budget
      // that decides between 2 and 3 waves per SIMD)
      constexpr int KSB = (FR == 1) ? BK / 4 : (FR == 2 ? BK / 8 : BK / 16);
#pragma unroll
      for (int kb = 0; kb < BK / 4; kb += KSB) {
        float af[KSB][FR], bf[KSB][FR];
#pragma unroll
        for (int ks = 0; ks < KSB; ++ks)
#pragma unroll
          for (int f = 0; f < FR; ++f) {
            af[ks][f] = As[((kb + ks) * 4 + q) * a_lk + (wm * 16 * FR + f * 16 + bi) * a_lm];
            bf[ks][f] = Bs[((kb + ks) * 4 + q) * b_lk + (wn * 16 * FR + f * 16 + bi) * b_ln];
          }
#pragma unroll
        for (int ks = 0; ks < KSB; ++ks)
#pragma unroll
          for (int fm = 0; fm < FR; ++fm)
#pragma unroll
            for (int fn = 0; fn < FR; ++fn) acc[fm][fn] = mma16x16x4(af[ks][fm], bf[ks][fn], acc[fm][fn]);
      }
      // next tile (kt+1, ring slot (s+1) % DEPTH) into the other image, then refill that slot with tile
      // kt+1+DEPTH; tiles beyond nkt are all-zero (clamped, masked loads): harmless
      const int sn = (s + 1) % DEPTH;
      store_tiles(sn, img ^ 1);
      load_tiles(sn, kbeg + (kt + 1 + DEPTH) * BK);
      lds_barrier();
    }
  }

  gemm_epilogue<FR, TF>(g, d, pi, z, split, m0, n0, wm, wn, bi, q, tid, lane, wave, acc, tgt, do_mse);
}

static int g_cus = 0;
int device_cus() {
  if (g_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      g_cus = prop.multiProcessorCount;
    if (g_cus <= 0) g_cus = 256;
  }
  return g_cus;
}

// Host-side launch of one group (count <= MFM_GEMM_MAXP).
int gemm_group_launch(const MfmGemmDesc* descs, int count, hipStream_t stream, const ZeroSpans* zs, const MseEpi* mse,
                      int mse_count, int precision, const GemmEpiSet* epis) {
  MFM_REQUIRE(count >= 1 && count <= MFM_GEMM_MAXP, "gemm group: count %d out of range", count);
  const int cus = device_cus();
  GemmGroup g;
  memset(&g, 0, sizeof(g));
  g.count = count;
  MFM_REQUIRE(mse_count >= 0 && mse_count <= 3 && mse_count <= count && (mse_count == 0 || mse), "gemm group: bad mse epilogue count %d", mse_count);
  g.mse_count = mse_count;
  for (int i = 0; i < mse_count; ++i) {
    MFM_REQUIRE(mse[i].x && !descs[i].accumulate && descs[i].batch == 1, "gemm group: mse epilogue %d needs a plain (non-accumulating, unbatched) product", i);
    g.mse[i] = mse[i];
  }
  if (epis && epis->count > 0) {
    MFM_REQUIRE(epis->count <= MFM_GEMM_NEPI && epis->count <= count && epis->epi, "gemm group: %d output transforms (max %d)", epis->count, MFM_GEMM_NEPI);
    for (int i = 0; i < epis->count; ++i) {
      const GemmEpi& e = epis->epi[i];
      MFM_REQUIRE(e.kind >= 0 && e.kind <= 3 && (e.kind == 0 || e.kind == 2 || e.aux), "gemm group: bad output transform %d", i);
      MFM_REQUIRE(e.kind == 0 || (!descs[i].accumulate && descs[i].split_k <= 1 && descs[i].batch == 1),
                  "gemm group: output transform %d needs a plain (non-accumulating, unbatched) product", i);
      g.epi[i] = e;
    }
    g.epi_count = epis->count; g.epi_train = epis->train; g.epi_seed = epis->seed; g.epi_tick = epis->tick;
  }
  if (zs) {
    for (int i = 0; i < MFM_GEMM_ZSPANS; ++i) {
      if (zs->n[i] <= 0) continue;
      MFM_REQUIRE((zs->n[i] & 3) == 0 && (((uintptr_t)zs->ptr[i]) & 15) == 0, "gemm group: zero span %d not 16-byte shaped", i);
      g.zero_ptr[i] = zs->ptr[i]; g.zero_n[i] = zs->n[i];
    }
  }
  // pass 1: block count with 64x64 tiles and no split; decide tile size
  long blocks64 = 0;
  for (int i = 0; i < count; ++i) {
    const MfmGemmDesc& d = descs[i];
    MFM_REQUIRE(d.m > 0 && d.n > 0 && d.k >= 0 && d.batch > 0, "gemm[%d]: bad dims m=%d n=%d k=%d batch=%d", i, d.m, d.n, d.k, d.batch);
    MFM_REQUIRE(d.a && d.b && (d.c || (i < mse_count && !d.accumulate && !d.c_bf16)), "gemm[%d]: null operand", i);
    if (d.a_bf16 || d.c_bf16) {
      MFM_REQUIRE(precision == 1, "gemm[%d]: bf16-resident operands (a_bf16 / c_bf16) are taken by the bf16 entry point only", i);
      MFM_REQUIRE(!d.c_bf16 || (!d.accumulate && d.split_k <= 1), "gemm[%d]: c_bf16 needs a plain (non-accumulating, unsplit) product", i);
      MFM_REQUIRE(!d.a_bf16 || ((((uintptr_t)d.a) & 15) == 0 && (d.a_sz & 7) == 0 &&
                                ((d.a_sk == 1 && (d.a_sm & 7) == 0) || (d.a_sm == 1 && d.a_sk != 1 && (d.a_sk & 7) == 0))),
                  "gemm[%d]: a_bf16 needs a unit-stride axis on 16-byte boundaries (a_sm=%lld a_sk=%lld a_sz=%lld)", i,
                  (long long)d.a_sm, (long long)d.a_sk, (long long)d.a_sz);
    }
    MFM_REQUIRE(d.n_valid >= 0 && d.n_valid <= d.n, "gemm[%d]: n_valid %d > n %d", i, d.n_valid, d.n);
    MFM_REQUIRE(d.split_k <= 1 || d.accumulate, "gemm[%d]: split_k needs accumulate", i);
    {
      const int64_t lim = (int64_t)1 << 31;
      const int64_t ea = (int64_t)(d.m - 1) * d.a_sm + (int64_t)(d.k > 0 ? d.k - 1 : 0) * d.a_sk;
      const int64_t eb = (int64_t)(d.k > 0 ? d.k - 1 : 0) * d.b_sk + (int64_t)(d.n_valid > 0 ? d.n_valid - 1 : 0) * d.b_sn;
      MFM_REQUIRE(ea < lim && eb < lim && ea >= 0 && eb >= 0, "gemm[%d]: operand spans >= 2^31 elements (32-bit tile offsets)", i);
    }
    blocks64 += (long)cdiv(d.m, 64) * cdiv(d.n, 64) * d.batch;
  }
  int FR = (blocks64 >= 2L * cus) ? 2 : 1;
  // 128x128 tiles (FR = 4) are opt-in: on single large forward-layout products they add ~10 % over 64x64 (fp32:
  // 61 -> 69 TF/s on the projection shape, 86 -> 99 TF/s at 4096^3), but the step's grouped launches mix in small-K
  // and weight-gradient problems that lose more than that (proj launch at B=2048: 457 -> 514 us), and the bf16-operand
  // kernel is faster at 64x64 throughout (profiles/r02_gemm_tiles.txt)
  if (const char* e = opt_get("MFM_GEMM_FR")) {      // tuning override
    FR = (e[0] == '4') ? 4 : ((e[0] == '2') ? 2 : 1);
    if (FR == 4 && mse_count > 0) FR = 2;
  }
  const int BT = 32 * FR;
  long base_blocks = 0;
  for (int i = 0; i < count; ++i)
    base_blocks += (long)cdiv(descs[i].m, BT) * cdiv(descs[i].n, BT) * descs[i].batch;
  long kdepth = 2048;      // K elements one workgroup walks at most (accumulating problems; measured 256..4096 at B=512/2048)
  if (const char* e = opt_get("MFM_GEMM_KDEPTH")) { const long v = atol(e); if (v >= 64) kdepth = v; }   // tuning override
  bool vec = true;
  for (int i = 0; i < count; ++i) {
    const MfmGemmDesc& d = descs[i];
    const bool a_mcontig = (d.a_sm == 1 && d.a_sk != 1), b_ncontig = (d.b_sn == 1 && d.b_sk != 1);
    if (!(a_mcontig || d.a_sk == 1) || !(b_ncontig || d.b_sk == 1)) vec = false;
  }
  // bf16 MFMA operands (precision 1) need every operand unit-stride along its 16-byte load groups; a group with an
  // oddly strided operand runs on the fp32 kernel's dword path instead
  const bool bf16 = (precision == 1) && vec;
  for (int i = 0; i < count; ++i)
    MFM_REQUIRE(bf16 || !(descs[i].a_bf16 || descs[i].c_bf16), "gemm[%d]: bf16-resident operands need unit-stride operands in the whole group", i);
  const int bk = bf16 ? BKB : BK;
  int total = 0;
  for (int i = 0; i < count; ++i) {
    GemmProblem& P = g.p[i];
    P.d = descs[i];
    P.tiles_m = cdiv(P.d.m, BT);
    P.tiles_n = cdiv(P.d.n, BT);
    int split = P.d.split_k;
    if (split <= 0) {
      // auto (accumulate problems only).  Two reasons to split K: (a) fill the chip when the whole
      // group has few blocks; (b) bound the serial K loop of ONE block -- a weight-gradient product
      // has a small output and K = T*B rows, and must not run as 10 blocks x 40960 deep just because
      // a large sibling problem already fills the grid.
      split = 1;
      if (P.d.accumulate && P.d.k >= 128) {
        const long want_fill = (3L * cus + base_blocks - 1) / base_blocks;
        const long want_depth = (P.d.k + kdepth - 1) / kdepth;
        long want = want_fill > want_depth ? want_fill : want_depth;
        const long maxs = P.d.k / 64;
        split = (int)(want < 1 ? 1 : (want > maxs ? maxs : want));
        if (split < 1) split = 1;
      }
    }
    int kps = round_up(cdiv(P.d.k > 0 ? P.d.k : 1, split), bk);
    split = cdiv(P.d.k > 0 ? P.d.k : 1, kps);
    P.d.split_k = split;
    P.k_per_split = kps;
    P.block_begin = total;
    g.begins[i] = total;
    total += P.tiles_m * P.tiles_n * P.d.batch * split;
  }
  for (int i = count; i < MFM_GEMM_MAXP; ++i) g.begins[i] = 0x7fffffff;
  if (bf16) return gemm_bf16_launch_kernel(g, FR, total, stream);
  if (FR == 4) {
    if (vec) MFM_LAUNCH_TIMED((gemm_f32_kernel<4, true>), dim3(total), dim3(256), 0, stream, g);
    else MFM_LAUNCH_TIMED((gemm_f32_kernel<4, false>), dim3(total), dim3(256), 0, stream, g);
  } else if (FR == 2) {
    if (vec) MFM_LAUNCH_TIMED((gemm_f32_kernel<2, true>), dim3(total), dim3(256), 0, stream, g);
    else MFM_LAUNCH_TIMED((gemm_f32_kernel<2, false>), dim3(total), dim3(256), 0, stream, g);
  } else {
    if (vec) MFM_LAUNCH_TIMED((gemm_f32_kernel<1, true>), dim3(total), dim3(256), 0, stream, g);
    else MFM_LAUNCH_TIMED((gemm_f32_kernel<1, false>), dim3(total), dim3(256), 0, stream, g);
  }
  MFM_LAUNCH_CHECK("gemm_f32_kernel");
  return MFM_OK;
}

}  // namespace mfm

extern "C" int mfm_gemm_grouped_f32(const MfmGemmDesc* descs, int count, void* stream) {
  if (!descs || count <= 0) {
    mfm::set_error("mfm_gemm_grouped_f32: no problems");
    return MFM_ERR_ARG;
  }
  for (int i = 0; i < count; ++i)       // (the library uses the reserved bytes internally: a caller's garbage there would be a device pointer)
    if (descs[i].reserved_[0] != 0 || descs[i].reserved_[1] != 0) {
      mfm::set_error("mfm_gemm_grouped_f32: problem %d has non-zero reserved_ fields (zero the struct before filling it)", i);
      return MFM_ERR_ARG;
    }
  int done = 0;
  while (done < count) {
    int n = count - done;
    if (n > MFM_GEMM_MAXP) n = MFM_GEMM_MAXP;
    int rc = mfm::gemm_group_launch(descs + done, n, (hipStream_t)stream);
    if (rc != MFM_OK) return rc;
    done += n;
  }
  return MFM_OK;
}

extern "C" int mfm_device_cus(void) { return mfm::device_cus(); }

extern "C" int mfm_gemm_grouped_bf16(const MfmGemmDesc* descs, int count, void* stream) {
  if (!descs || count <= 0) {
    mfm::set_error("mfm_gemm_grouped_bf16: no problems");
    return MFM_ERR_ARG;
  }
  for (int i = 0; i < count; ++i)
    if (descs[i].reserved_[0] != 0 || descs[i].reserved_[1] != 0) {
      mfm::set_error("mfm_gemm_grouped_bf16: problem %d has non-zero reserved_ fields (zero the struct before filling it)", i);
      return MFM_ERR_ARG;
    }
  int done = 0;
  while (done < count) {
    int n = count - done;
    if (n > MFM_GEMM_MAXP) n = MFM_GEMM_MAXP;
    int rc = mfm::gemm_group_launch(descs + done, n, (hipStream_t)stream, nullptr, nullptr, 0, 1);
    if (rc != MFM_OK) return rc;
    done += n;
  }
  return MFM_OK;
}
