// Weight gradients of the LSTMs at large T*B, ONE pass over dA.
//
// Reference: what loss.backward() accumulates into weight_ih / weight_hh / bias_ih / bias_hh of every nn.LSTMCell
// (mfm_model.py:14-91 unrolled over T).  With A_t = the gate pre-activation gradients [B, 4h] the BPTT left in the
// gates buffer:
//     dW_ih = sum_t A_t^T x_t          dW_hh = sum_{t>=1} A_t^T h_{t-1}          db_ih = db_hh = sum_t A_t^T 1
// (decoders: the step input for t >= 1 IS h_{t-1}, so the recurrent sum goes to both matrices; their t = 0 term with
// h_init stays in the tail GEMM).  As grouped TN GEMMs these are 3 problems per LSTM on 32x32 tiles: at B=2048 the
// launch reads 1.37 GB for 0.4 GB of operands and every 8 KB of tile loads feeds 64 K flops (profiles/r02_roofline_table_
// l_fp32.txt) -- a CU sustains ~14 B/clk from L2, so the tile shape, not the MFMA, sets the speed.
//
// Here the three sums of one LSTM are ONE product C[4 Hp, N] += A^T [x | h_prev | 1] over all rows: a workgroup owns 96
// columns of A and ALL N = d + h + 1 columns of the right-hand side for a contiguous range
// of rows, so every row of [x | h_prev] is fetched once per 96 A-columns and A once: ~40 flops per loaded byte
// instead of 8.  Rows are walked in chunks (32 rows bf16 / 16 rows fp32): global -> registers (one chunk ahead) -> LDS,
// transposed to [column][row] on the way (4 x 4 register blocks) so that MFMA fragments are contiguous LDS reads;
// row ranges (split-K) fill the chip and are combined with atomics into the gradient buffer.
//
// STATUS: parity-green (tests/test_gpu_large_batch.py "dwonepass", bf16 and MFN variants) but SLOWER than the grouped GEMMs it
// replaces at B=2048 (profiles/r02_dw_onepass.txt: fp32 772 + 93 us against 635, bf16 452 + 67 against 372), so it is
// OPT-IN (MFM_DW_ONEPASS_MINROWS=<rows>).  Measured causes: (a) one burst of 12-16 loads per thread and chunk, waited for
// at the top of the next chunk, sustains only ~5.6 B/clk per CU from L2 -- a second chunk in flight needs 48-64 more
// staging registers next to 108 accumulators; (b) the per-fragment `nf < NF` guards compile to exec-mask branches with
// one LDS read + wait per three MFMAs (~50 % MFMA rate in fp32).
//
// 512 threads = 8 waves as 2 (A fragments 3 each) x 4 (N fragments, strided, <= 9 each): 27 accumulator tiles per wave.
// bf16 plans: operands rounded to bf16 on the way into LDS, v_mfma_f32_16x16x32_bf16; fp32 plans: v_mfma_f32_16x16x4_f32.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "internal.h"

namespace mfm {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int DW_THREADS = 512;
constexpr int DW_MF = 3;                     // A fragments per wave (4 would need 144 accumulator registers + 64 of staging: spills)
constexpr int DW_MT = 2 * 16 * DW_MF;        // A columns per workgroup: 96
constexpr int DW_NFW = 9;                    // N fragments per wave: N (padded) <= 16 * 4 * 9 = 576
constexpr int DW_OOB = 0x7FFFFFF0;

template <bool BF16> struct DwCfg;
template <> struct DwCfg<true> { static constexpr int KC = 32, XB = 2, HB = 1; };     // blocks (4 rows x 4 columns) per thread
template <> struct DwCfg<false> { static constexpr int KC = 16, XB = 1, HB = 1; };

template <bool BF16>
__global__ __launch_bounds__(DW_THREADS) void dw_onepass_kernel(const DwLaunch L) {
  using Cfg = DwCfg<BF16>;
  constexpr int KC = Cfg::KC, RG = KC / 4, XB = Cfg::XB, HB = Cfg::HB;
  constexpr int LDK = KC + 8 / (BF16 ? 1 : 2);            // LDS row stride in elements: 80 bytes in both precisions
  using elem_t = typename std::conditional<BF16, __bf16, float>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];

  // ---- which item / A tile / row range (wave-uniform)
  int it = 0;
#pragma unroll
  for (int i = 1; i < MFM_DW_MAXI; ++i) it += (i < L.n_items && (int)blockIdx.x >= L.it[i].tile_begin) ? 1 : 0;
  const DwItem& I = L.it[it];
  const int local = (int)blockIdx.x - I.tile_begin;
  const int mt = local % I.m_tiles, sp = local / I.m_tiles;
  const int m0 = mt * DW_MT;
  const int r_begin = sp * I.rows_per_split, r_end = min(L.rows, r_begin + I.rows_per_split);
  const int px = (I.dx + 3) & ~3, ph = (I.hN + 3) & ~3;     // segment extents in the concatenated right-hand side
  const int NP = (px + ph + 4 + 15) & ~15;
  const int NF = NP >> 4;
  elem_t* At = reinterpret_cast<elem_t*>(dsm);               // [2][DW_MT][LDK]
  elem_t* Bt = At + 2 * DW_MT * LDK;                          // [2][NP][LDK]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bi = lane & 15, q = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;

  // ---- load plan: block = 4 rows x 4 columns; thread -> (row group rg, column group) per operand
  const int rg = tid % RG, cgi = tid / RG;                    // cgi < 512 / RG
  constexpr int CGS = DW_THREADS / RG;                         // column groups covered per round
  const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc((void*)I.dA, 0, (int)min((int64_t)L.rows * I.ldA * 4, (int64_t)0x7FFFFFF0), 0x00020000);
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)(I.x ? I.x : I.dA), 0, I.x ? (int)min(((int64_t)(L.rows - 1) * I.ldx + I.dx) * 4, (int64_t)0x7FFFFFF0) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t hres = __builtin_amdgcn_make_buffer_rsrc((void*)I.hs, 0, (int)min((int64_t)L.rows * I.ldh * 4, (int64_t)0x7FFFFFF0), 0x00020000);
  const int ldA = I.ldA, ldx = (int)I.ldx, ldh = I.ldh, dx = I.dx, hN = I.hN, shift = I.shift, M = I.M;

  f32x4 ra[4], rx[XB][4], rh[HB][4];
  auto request = [&](int r0) {
    // rows r0 + 4 rg + {0..3}; rows >= r_end are requested out of range (zeros)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = r0 + 4 * rg + e;
      const bool rok = r < r_end;
      {
        const int col = m0 + 4 * cgi;
        const bool ok = (int)rok & (int)(cgi < DW_MT / 4) & (int)(col < M);
        ra[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ares, ok ? (r * ldA + col) * 4 : DW_OOB, 0, 0));
      }
#pragma unroll
      for (int b = 0; b < XB; ++b) {
        const int j = 4 * (cgi + b * CGS);
        const bool ok = (int)rok & (int)(j < dx);
        rx[b][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, ok ? (r * ldx + j) * 4 : DW_OOB, 0, 0));
      }
#pragma unroll
      for (int b = 0; b < HB; ++b) {
        const int j = 4 * (cgi + b * CGS);
        const bool ok = (int)rok & (int)(j < hN) & (int)(r >= shift);
        rh[b][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hres, ok ? ((r - shift) * ldh + j) * 4 : DW_OOB, 0, 0));
      }
    }
  };
  // 4 x 4 register block (rows e, columns c) -> LDS [column][row]; elements past `nvalid` columns of the group's row
  // belong to the next row of the operand (or another tensor): multiplied away
  auto park = [&](elem_t* tile, int col0, f32x4 (&blk)[4], int nvalid) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float keep = (c < nvalid) ? 1.0f : 0.0f;
      f32x4 v = f32x4{blk[0][c], blk[1][c], blk[2][c], blk[3][c]} * keep;
      if constexpr (BF16) *reinterpret_cast<bf16x4*>(tile + (col0 + c) * LDK + 4 * rg) = __builtin_convertvector(v, bf16x4);
      else *reinterpret_cast<f32x4*>(tile + (col0 + c) * LDK + 4 * rg) = v;
    }
  };
  auto store = [&](int buf, int r0) {
    elem_t* A = At + buf * DW_MT * LDK;
    elem_t* B = Bt + buf * NP * LDK;
    if (cgi < DW_MT / 4) park(A, 4 * cgi, ra, 4);
#pragma unroll
    for (int b = 0; b < XB; ++b) {
      const int j = 4 * (cgi + b * CGS);
      if (j < px) park(B, j, rx[b], dx - j);
    }
#pragma unroll
    for (int b = 0; b < HB; ++b) {
      const int j = 4 * (cgi + b * CGS);
      if (j < ph) park(B, px + j, rh[b], hN - j);
    }
    // the ones column (bias gradients) and the padding columns behind it
    const int npad = NP - (px + ph);
    if (tid < npad * RG) {
      const int c = tid / RG, g = tid % RG;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (c == 0 && r0 + 4 * g + e < r_end) ? 1.0f : 0.0f;
      if constexpr (BF16) *reinterpret_cast<bf16x4*>(B + (px + ph + c) * LDK + 4 * g) = __builtin_convertvector(v, bf16x4);
      else *reinterpret_cast<f32x4*>(B + (px + ph + c) * LDK + 4 * g) = v;
    }
  };

  f32x4 acc[DW_MF][DW_NFW];
#pragma unroll
  for (int i = 0; i < DW_MF; ++i)
#pragma unroll
    for (int j = 0; j < DW_NFW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  request(r_begin);
  int buf = 0;
  for (int r0 = r_begin; r0 < r_end; r0 += KC) {
    store(buf, r0);
    lds_barrier();
    request(r0 + KC);                         // unconditional: rows >= r_end come back as zeros (no load under a branch)
    const elem_t* A = At + buf * DW_MT * LDK + (wm * 16 * DW_MF + bi) * LDK;
    const elem_t* B = Bt + buf * NP * LDK + bi * LDK;
    if constexpr (BF16) {
      bf16x8 af[DW_MF];
#pragma unroll
      for (int i = 0; i < DW_MF; ++i) af[i] = *reinterpret_cast<const bf16x8*>(A + i * 16 * LDK + 8 * q);
#pragma unroll
      for (int j = 0; j < DW_NFW; ++j) {
        const int nf = wn + 4 * j;
        if (nf < NF) {
          const bf16x8 bf = *reinterpret_cast<const bf16x8*>(B + nf * 16 * LDK + 8 * q);
#pragma unroll
          for (int i = 0; i < DW_MF; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf, acc[i][j], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < KC / 4; ++ks) {
        float af[DW_MF];
#pragma unroll
        for (int i = 0; i < DW_MF; ++i) af[i] = A[i * 16 * LDK + 4 * ks + q];
#pragma unroll
        for (int j = 0; j < DW_NFW; ++j) {
          const int nf = wn + 4 * j;
          if (nf < NF) {
            const float bf = B[nf * 16 * LDK + 4 * ks + q];
#pragma unroll
            for (int i = 0; i < DW_MF; ++i) acc[i][j] = mma16x16x4(af[i], bf, acc[i][j]);
          }
        }
      }
    }
    buf ^= 1;
  }

  // ---- add the tile into the gradient buffers: A column m -> (gate, unit), N column -> (segment, j)
  const int Hp = I.Hp, h = I.h;
#pragma unroll
  for (int j = 0; j < DW_NFW; ++j) {
    const int nf = wn + 4 * j;
    if (nf >= NF) continue;
    const int n = nf * 16 + bi;
    float* dst = nullptr; float* dst2 = nullptr;
    int ldc = 0, col = 0;
    if (n < px) { if (n < dx) { dst = I.c_x; ldc = I.ldc_x; col = n; } }
    else if (n < px + ph) { if (n - px < hN) { dst = I.c_h; dst2 = I.c_h2; ldc = I.ldc_h; col = n - px; } }
    else if (n == px + ph) { dst = I.c_b; dst2 = I.c_b2; ldc = 1; col = 0; }
    if (!dst) continue;
#pragma unroll
    for (int i = 0; i < DW_MF; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 16 * DW_MF + i * 16 + 4 * q + r;
        if (m >= M) continue;
        const int g = m / Hp, u = m - g * Hp;
        if (u >= h) continue;
        const int64_t o = (int64_t)(g * h + u) * ldc + col;
        atomicAdd(dst + o, acc[i][j][r]);
        if (dst2) atomicAdd(dst2 + o, acc[i][j][r]);
      }
  }
}

}  // namespace

// MFM_ERR_UNSUPPORTED: an item this kernel does not take (the caller keeps the GEMM problems)
int dw_onepass_supported(const DwItem& I, int precision) {
  const int px = (I.dx + 3) & ~3, ph = (I.hN + 3) & ~3;
  const int NP = (px + ph + 4 + 15) & ~15;
  const int RG = (precision ? 32 : 16) / 4, CGS = DW_THREADS / RG;
  if (NP > 16 * 4 * DW_NFW) return 0;
  if (px > 4 * CGS * (precision ? 2 : 1) || ph > 4 * CGS) return 0;
  if ((I.M & 15) || I.Hp < I.h || (I.Hp & 15)) return 0;
  return 1;
}

int dw_onepass_launch(DwLaunch& L, int precision, hipStream_t stream) {
  MFM_REQUIRE(L.n_items >= 1 && L.n_items <= MFM_DW_MAXI && L.rows >= 1, "dw onepass: bad launch");
  const int KC = precision ? 32 : 16;
  const int LDK = precision ? 40 : 20;
  const size_t esz = precision ? 2 : 4;
  // Row ranges (split-K): a chunk of rows costs a workgroup a fixed part (load round trip, barrier) plus a part
  // proportional to N, so row ranges are sized by (1 + N / 128) -- with ranges proportional to N alone the 25-column
  // decoders walked 20 k rows in 2 workgroups and set the launch time (profiles/r02_dw_onepass.txt)
  double wsum = 0.0;
  for (int i = 0; i < L.n_items; ++i) {
    DwItem& I = L.it[i];
    MFM_REQUIRE(I.dA && I.hs && I.c_h && I.c_b && dw_onepass_supported(I, precision), "dw onepass: item %d", i);
    MFM_REQUIRE((int64_t)L.rows * I.ldA < ((int64_t)1 << 29) && (int64_t)L.rows * std::max<int64_t>(I.ldx, I.ldh) < ((int64_t)1 << 29),
                "dw onepass: operand spans >= 2^31 bytes");
    I.m_tiles = (I.M + DW_MT - 1) / DW_MT;
    wsum += (double)I.m_tiles * (1.0 + (I.dx + I.hN + 1) / 128.0);
  }
  double target = 4.0 * device_cus();
  if (const char* e = opt_get("MFM_DW_TARGET")) target = atof(e) * device_cus();

  int tiles = 0;
  size_t smem = 0;
  for (int i = 0; i < L.n_items; ++i) {
    DwItem& I = L.it[i];
    const int N = I.dx + I.hN + 1;
    int splits = (int)(target * (1.0 + N / 128.0) / wsum + 0.5);
    const int max_splits = std::max(1, L.rows / (4 * KC));
    splits = std::max(1, std::min(splits, max_splits));
    I.rows_per_split = ((L.rows + splits - 1) / splits + KC - 1) / KC * KC;
    I.splits = (L.rows + I.rows_per_split - 1) / I.rows_per_split;
    I.tile_begin = tiles;
    tiles += I.m_tiles * I.splits;
    const int px = (I.dx + 3) & ~3, ph = (I.hN + 3) & ~3;
    const int NP = (px + ph + 4 + 15) & ~15;
    smem = std::max(smem, (size_t)2 * (DW_MT + NP) * LDK * esz);
  }
  static bool attr = false;
  if (!attr) {
    MFM_HIP_CHECK(hipFuncSetAttribute((const void*)dw_onepass_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    MFM_HIP_CHECK(hipFuncSetAttribute((const void*)dw_onepass_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    attr = true;
  }
  MFM_REQUIRE(smem <= 128 * 1024, "dw onepass: %zu bytes of LDS", smem);
  if (precision) hipLaunchKernelGGL(dw_onepass_kernel<true>, dim3(tiles), dim3(DW_THREADS), smem, stream, L);
  else hipLaunchKernelGGL(dw_onepass_kernel<false>, dim3(tiles), dim3(DW_THREADS), smem, stream, L);
  MFM_LAUNCH_CHECK("dw_onepass_kernel");
  return MFM_OK;
}

}  // namespace mfm
