// Grouped, batched, strided GEMM with bf16 MFMA operands and fp32 accumulation (gfx950):
// v_mfma_f32_16x16x32_bf16, 16x the rate of the f32-input instruction the fp32 path uses.
//
// Same contract and problem table as gemm.hip (MfmGemmDesc, up to MFM_GEMM_MAXP problems per launch, the same
// epilogue), and the operands are the SAME fp32 buffers: the fp32 -> bf16 rounding (nearest even,
// v_cvt_pk_bf16_f32) happens in registers on the way from the global tile to its LDS image, so master
// weights, saved activations and gradients stay fp32 in HBM ("bf16 compute", BASELINE.json configs 2-4) and a
// plan switches precision without a second copy of anything.
//
// A bf16 MFMA fragment is 8 consecutive k of one row, so both LDS images are [row][k] with k contiguous
// (one ds_read_b128 per fragment; rows padded by 16 bytes so that the 16 rows of a fragment read fall on
// different banks).  Operands that are k-contiguous in memory (x, h, W of a forward product) are written with
// one ds_write_b64 per 16-byte load.  Operands that are contiguous along m / n instead -- BOTH operands of every
// weight-gradient product dW = dA^T X, where k runs over the T*B rows -- are transposed for free in registers:
// a thread fetches a 4 (m) x NK (k) block with NK 16-byte loads along m and writes NK bf16 per row.
#include <stdlib.h>

#include "gemm_common.h"

namespace mfm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int FR>
__global__ __launch_bounds__(256, (FR == 1) ? 4 : 2) void gemm_bf16_kernel(const GemmGroup g) {
  static_assert(FR == 1 || FR == 2 || FR == 4, "tile sizes 32, 64, 128");
  constexpr int BM = 32 * FR, BN = 32 * FR;
  constexpr int LDK = BKB + 8;                       // bf16 elements per LDS row
  constexpr int EPT = BM * BKB / 256;                // fp32 elements per thread and operand tile: 8 / 16
  constexpr int G = EPT / 4;                         // 16-byte loads per thread and operand tile: 2 / 4 (= NK below)
  __shared__ __attribute__((aligned(16))) __bf16 As2[2][BM * LDK];
  __shared__ __attribute__((aligned(16))) __bf16 Bs2[2][BN * LDK];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int bi = lane & 15, q = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- locate problem / tile (wave-uniform); XCD-aware tile order as in gemm.hip
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

  // a_bf16: the A operand is a bf16-RESIDENT buffer (saved hidden states / gate gradients of a bf16 plan): element size 2,
  // no rounding pass -- the 16-byte loads go to the LDS image as they are (wave-uniform per problem)
  const bool a16 = d.a_bf16 != 0;
  const float* __restrict__ A = a16 ? reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(d.a) + (int64_t)z * d.a_sz)
                                    : d.a + (int64_t)z * d.a_sz;
  const float* __restrict__ Bm = d.b + (int64_t)z * d.b_sz;
  const bool a_mcontig = (d.a_sm == 1 && d.a_sk != 1);
  const bool b_ncontig = (d.b_sn == 1 && d.b_sk != 1);
  const int a_sm = (int)d.a_sm, a_sk = (int)d.a_sk, b_sk = (int)d.b_sk, b_sn = (int)d.b_sn;
  // (bf16: rounded up to whole dwords -- the range check is per dword, an odd element count would zero the last element)
  const int a_bytes = a16 ? ((((d.m - 1) * a_sm + (max(d.k, 1) - 1) * a_sk + 1) * 2 + 3) & ~3)
                          : ((d.m - 1) * a_sm + (max(d.k, 1) - 1) * a_sk + 1) * 4;
  const int b_bytes = ((max(d.k, 1) - 1) * b_sk + (max(d.n_valid, 1) - 1) * b_sn + 1) * 4;
  const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t bres = __builtin_amdgcn_make_buffer_rsrc((void*)Bm, 0, b_bytes, 0x00020000);
  const int klast = max(kend - 1, 0);

  // Per-thread group coordinates inside an operand tile [rows][BKB].
  //   k-contiguous operand: G groups (row, k..k+3), consecutive groups of a thread are consecutive in k.
  //   row-contiguous operand: ONE block of 4 rows x G consecutive k (G loads of 16 bytes along the rows).
  struct Coord { int row, k; };
  auto coord = [&](bool rowcontig, int rows, int gidx) -> Coord {
    Coord c;
    if (rowcontig) {
      const int rb = tid % (rows / 4), kb = tid / (rows / 4);
      c.row = 4 * rb; c.k = G * kb + gidx;
    } else {
      const int idx = (tid * G + gidx) * 4;
      c.row = idx / BKB; c.k = idx % BKB;
    }
    return c;
  };
  f32x4 ra[G], rb[G];
  auto load_op = [&](f32x4 (&r)[G], const __amdgpu_buffer_rsrc_t res, bool rowcontig, int rows, int r0, int rmax,
                     int s_row, int s_k, int k0) {
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
      const Coord c = coord(rowcontig, rows, gi);
      const int gr = r0 + c.row, gk = k0 + c.k;
      const int off = min(gr, rmax) * s_row + min(gk, klast) * s_k;
      const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res, off * 4, 0, 0));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int er = rowcontig ? gr + e : gr, ek = rowcontig ? gk : gk + e;
        r[gi][e] = v[e] * (float)((int)(er <= rmax) & (int)(ek < kend));
      }
    }
  };
  auto store_op = [&](const f32x4 (&r)[G], __bf16* img, bool rowcontig, int rows) {
    if (rowcontig) {
      const Coord c = coord(true, rows, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {          // row c.row + e receives k = c.k .. c.k + G - 1
        __bf16* p = img + (c.row + e) * LDK + c.k;
        if constexpr (G == 2) {
          const f32x2 t = {r[0][e], r[1][e]};
          *reinterpret_cast<bf16x2*>(p) = __builtin_convertvector(t, bf16x2);
        } else if constexpr (G == 4) {
          const f32x4 t = {r[0][e], r[1][e], r[2][e], r[3][e]};
          *reinterpret_cast<bf16x4*>(p) = __builtin_convertvector(t, bf16x4);
        } else {
          const f32x4 t0 = {r[0][e], r[1][e], r[2][e], r[3][e]}, t1 = {r[4][e], r[5][e], r[6][e], r[7][e]};
          const bf16x4 b0 = __builtin_convertvector(t0, bf16x4), b1 = __builtin_convertvector(t1, bf16x4);
          *reinterpret_cast<bf16x8*>(p) = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      }
    } else {
#pragma unroll
      for (int gi = 0; gi < G; ++gi) {
        const Coord c = coord(false, rows, gi);
        *reinterpret_cast<bf16x4*>(img + c.row * LDK + c.k) = __builtin_convertvector(r[gi], bf16x4);
      }
    }
  };
  const int a_rmax = d.m - 1, b_rmax = max(d.n_valid - 1, 0);
  // ---- bf16-resident A: GA = FR 16-byte loads of 8 bf16 per thread and tile, kept as raw bits in ra[0 .. GA)
  constexpr int GA = FR;
  static_assert(GA <= G, "the bf16-resident loads reuse the fp32 staging registers");
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  //   k-contiguous (hs of a forward product): (row, 8 consecutive k) -> one ds_write_b128 into the [row][k] image
  //   m-contiguous (dA of a weight-gradient product; only small leftovers come here -- the LSTM sums over the rows have
  //   their own kernel, dw_bf16.hip): (k, 8 consecutive rows) -> eight 2-byte writes
  auto load_a16 = [&](int k0) {
#pragma unroll
    for (int gi = 0; gi < GA; ++gi) {
      const int l = tid + gi * 256;
      int row, k;
      if (a_mcontig) { k = l / (BM / 8); row = (l % (BM / 8)) * 8; }
      else { row = l / (BKB / 8); k = (l % (BKB / 8)) * 8; }
      const int gr = m0 + row, gk = k0 + k;
      const int off = min(gr, a_rmax) * a_sm + min(gk, klast) * a_sk;
      u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ares, off * 2, 0, 0));
#pragma unroll
      for (int e = 0; e < 4; ++e) {       // dword e holds elements 2e, 2e+1 along the contiguous axis
        const int r0 = a_mcontig ? gr + 2 * e : gr, r1 = a_mcontig ? gr + 2 * e + 1 : gr;
        const int c0 = a_mcontig ? gk : gk + 2 * e, c1 = a_mcontig ? gk : gk + 2 * e + 1;
        const unsigned m0k = ((int)(r0 <= a_rmax) & (int)(c0 < kend)) ? 0x0000FFFFu : 0u;
        const unsigned m1k = ((int)(r1 <= a_rmax) & (int)(c1 < kend)) ? 0xFFFF0000u : 0u;
        v[e] &= (m0k | m1k);
      }
      ra[gi] = __builtin_bit_cast(f32x4, v);
    }
  };
  auto store_a16 = [&](__bf16* img) {
#pragma unroll
    for (int gi = 0; gi < GA; ++gi) {
      const int l = tid + gi * 256;
      if (a_mcontig) {
        const int k = l / (BM / 8), row = (l % (BM / 8)) * 8;
        const u32x4 v = __builtin_bit_cast(u32x4, ra[gi]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned short bits = (unsigned short)((v[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu);
          *reinterpret_cast<unsigned short*>(img + (row + e) * LDK + k) = bits;
        }
      } else {
        const int row = l / (BKB / 8), k = (l % (BKB / 8)) * 8;
        *reinterpret_cast<f32x4*>(img + row * LDK + k) = ra[gi];
      }
    }
  };
  auto load_tiles = [&](int k0) {
    if (a16) load_a16(k0);
    else load_op(ra, ares, a_mcontig, BM, m0, a_rmax, a_sm, a_sk, k0);
    load_op(rb, bres, b_ncontig, BN, n0, b_rmax, b_sn, b_sk, k0);
  };
  auto store_tiles = [&](int img) {
    if (a16) store_a16(As2[img]);
    else store_op(ra, As2[img], a_mcontig, BM);
    store_op(rb, Bs2[img], b_ncontig, BN);
  };

  // squared-error epilogue: the targets of this thread's outputs are requested before the K loop
  const bool do_mse = (FR <= 2) && pi < g.mse_count;          // wave-uniform
  const MseEpi& me = g.mse[do_mse ? pi : 0];
  constexpr int TF = (FR <= 2) ? FR : 1;
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

  const int nkt = (kend - kbeg + BKB - 1) / BKB;
  load_tiles(kbeg);
  store_tiles(0);
  load_tiles(kbeg + BKB);              // tiles past the end are clamped + masked to zero: no branch around a load
  lds_barrier();
  for (int kt = 0; kt < nkt; ++kt) {
    const int img = kt & 1;
    const __bf16* As = As2[img] + (wm * 16 * FR + bi) * LDK + 8 * q;
    const __bf16* Bs = Bs2[img] + (wn * 16 * FR + bi) * LDK + 8 * q;
    bf16x8 af[2][FR], bf[2][FR];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int f = 0; f < FR; ++f) {
        af[ks][f] = *reinterpret_cast<const bf16x8*>(As + f * 16 * LDK + ks * 32);
        bf[ks][f] = *reinterpret_cast<const bf16x8*>(Bs + f * 16 * LDK + ks * 32);
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int fm = 0; fm < FR; ++fm)
#pragma unroll
        for (int fn = 0; fn < FR; ++fn)
          acc[fm][fn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][fm], bf[ks][fn], acc[fm][fn], 0, 0, 0);
    store_tiles(img ^ 1);                      // tile kt+1 (in registers since the previous iteration)
    load_tiles(kbeg + (kt + 2) * BKB);
    lds_barrier();
  }

  gemm_epilogue<FR, TF>(g, d, pi, z, split, m0, n0, wm, wn, bi, q, tid, lane, wave, acc, tgt, do_mse);
}

int gemm_bf16_launch_kernel(const GemmGroup& g, int FR, int total, hipStream_t stream) {
  if (FR == 4) MFM_LAUNCH_TIMED((gemm_bf16_kernel<4>), dim3(total), dim3(256), 0, stream, g);
  else if (FR == 2) MFM_LAUNCH_TIMED((gemm_bf16_kernel<2>), dim3(total), dim3(256), 0, stream, g);
  else MFM_LAUNCH_TIMED((gemm_bf16_kernel<1>), dim3(total), dim3(256), 0, stream, g);
  MFM_LAUNCH_CHECK("gemm_bf16_kernel");
  return MFM_OK;
}

}  // namespace mfm
