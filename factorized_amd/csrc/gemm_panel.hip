// Row-panel GEMM for the input projections at LARGE batch (round 2): one workgroup keeps a panel of A rows -- the
// input batch x[t,b,:] for 64-80 (fp32) / 96-160 (bf16) rows, ALL D feature columns -- resident in LDS and walks every
// output column of every encoder that consumes it (x W_ih^T + b_ih + b_hh for the four gates of the early-fusion
// encoder, the three modality encoders and, for the MFN variants, the three MFN LSTMs), streaming the weight tiles
// from L2.
//
// Why: the tiled kernels (gemm.hip, gemm_bf16.hip) are bound by the per-CU vector-memory load rate (~11 B/clk), not
// by the matrix pipes: a 64x64 tile loads 16 FLOP per byte, so the fp32 kernel tops out at 55-60 % of the MFMA peak
// and the bf16-operand kernel (whose matrix work is 16x cheaper) at whatever 1.9 GB of tile loads take for 211 MB
// of operands.  Here x is fetched from HBM exactly once (each encoder addresses its own column range of the
// shared panel: the l / a / v slices are column ranges of the same rows), and the only repeated traffic is the
// weight set (0.8 MB, L2-resident) once per panel: 28 (fp32) / 56 (bf16) FLOP per loaded byte.
//
// Column groups: group = one LSTM: (weight block [4h, k_len] row-major with row stride ldw, column offset k_off into the
// panel, biases, C block [., 4 Hp] with per-gate padding -- internal.h::PanelGroup).  K-tiles are taken on the PANEL's tile grid (multiples of BK from column 0), so the A fragments
// are always aligned; a weight tile is addressed at (k - k_off) with 4-byte-aligned 16-byte buffer loads and masked
// to [0, k_len): only the tiles that overlap a group's column range are visited.
#include <stdlib.h>

#include <type_traits>

#include "internal.h"

namespace mfm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int PANEL_BN = 128;
constexpr int PANEL_THREADS = 512;

// Panel height.  The 8 waves form a WM x WN grid, each wave owning FM x FN 16x16 fragments: BM = 16 WM FM rows of the
// panel, 16 WN FN = 128 columns of a job.  One workgroup per CU (the panel fills the LDS), so a launch runs in
// ceil(panels / CUs) rounds and a round costs a + b BM (a: the 1.3 MB weight set streamed from L2 once per panel plus
// the fixed parts, 59 rows' worth in fp32 and 128 in bf16; profiles/r02_gemm_panel.txt).  The launcher therefore
// picks the height that minimises rounds x (a + BM): at T*B = 40960 that is 80 rows / 2 rounds in fp32 (64 rows took
// 2.5 -> 3) and 160 rows / 1 round in bf16 (128 rows took 1.25 -> 2).
template <bool BF16, int WM, int WN, int FM, int FN>
__global__ __launch_bounds__(PANEL_THREADS) void gemm_panel_kernel(const PanelLaunch L) {
  static_assert(WM * WN == PANEL_THREADS / 64 && WN * FN * 16 == PANEL_BN, "wave grid must cover 8 waves x 128 columns");
  constexpr int BM = 16 * WM * FM, BK = BF16 ? 64 : 32;
  constexpr int BN = PANEL_BN;
  constexpr int LDB = BK + (BF16 ? 8 : 4);            // elements per B-tile row in LDS
  constexpr int GB = BN * BK / 4 / PANEL_THREADS;      // 16-byte weight loads per thread and tile: 4 (bf16) / 2 (fp32)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int KT = (L.K + BK - 1) / BK;
  const int LDA = KT * BK + (BF16 ? 8 : 4);            // elements per panel row
  using elem_t = typename std::conditional<BF16, __bf16, float>::type;
  elem_t* Ap = reinterpret_cast<elem_t*>(smem);                                   // [BM][LDA]
  elem_t* Bs = Ap + (size_t)BM * LDA;                                             // [2][BN][LDB]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int bi = lane & 15, q = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * BM;

  // ---- the A panel: rows m0 .. m0+BM-1, columns 0 .. K-1 (zero beyond), fetched once
  {
    const int a_bytes = (int)(((int64_t)(L.M - 1) * L.lda + L.K) * 4);
    const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc((void*)L.a, 0, a_bytes, 0x00020000);
    const int groups_per_row = KT * BK / 4;
    const int total = BM * groups_per_row;
    constexpr int U = 8;                          // loads in flight per thread (a serial loop pays one HBM round trip per group)
    for (int base = tid; base < total; base += PANEL_THREADS * U) {
      f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = min(base + u * PANEL_THREADS, total - 1);
        const int r = idx / groups_per_row, k = (idx - r * groups_per_row) * 4;
        const int off = (int)(((int64_t)min(m0 + r, L.M - 1) * L.lda + min(k, L.K - 1)) * 4);
        v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ares, off, 0, 0));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = base + u * PANEL_THREADS;
        if (idx < total) {
          const int r = idx / groups_per_row, k = (idx - r * groups_per_row) * 4;
          f32x4 w = v[u];
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = (m0 + r < L.M && k + e < L.K) ? w[e] : 0.0f;
          if constexpr (BF16) {
            *reinterpret_cast<bf16x4*>(Ap + (size_t)r * LDA + k) = __builtin_convertvector(w, bf16x4);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) Ap[(size_t)r * LDA + k + e] = w[e];
          }
        }
      }
    }
  }

  // one "job" = (group, 128-column chunk); every job walks the K-tiles that overlap the group's column range
  f32x4 breg[GB];
  auto load_b = [&](const PanelGroup& G, const __amdgpu_buffer_rsrc_t wres, int n0, int kt) {
    // a group whose column range does not start on a multiple of 4 needs elements BEFORE its first 16-byte group: a
    // 16-byte load there would start in front of the weight block (a negative offset: the whole load reads 0, and the
    // compiler merges four consecutive dword loads back into exactly that), so such groups (the 5- and 20-column
    // modality slices) take dword loads at offsets clamped to the block and masked afterwards
    const bool dwords = (G.k_off & 3) != 0;        // wave-uniform
#pragma unroll
    for (int j = 0; j < GB; ++j) {
      const int idx = tid + j * PANEL_THREADS;
      const int nn = idx / (BK / 4), kk = (idx % (BK / 4)) * 4;
      const int n = n0 + nn;
      const int sg = n / G.seg, u = n - sg * G.seg;
      const bool nok = n < G.n && u < G.seg_valid;
      const int wrow = nok ? sg * G.seg_valid + u : 0;
      const int ko = kt * BK + kk - G.k_off;                         // column inside the weight block
      f32x4 v;
      if (dwords) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wres, (int)(((int64_t)wrow * G.ldw + max(ko + e, 0)) * 4), 0, 0));
      } else {
        v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, (int)(((int64_t)wrow * G.ldw + max(ko, 0)) * 4), 0, 0));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= (float)((int)nok & (int)(ko + e >= 0) & (int)(ko + e < G.k_len));
      breg[j] = v;
    }
  };
  auto store_b = [&](int buf) {
    elem_t* B = Bs + (size_t)buf * BN * LDB;
#pragma unroll
    for (int j = 0; j < GB; ++j) {
      const int idx = tid + j * PANEL_THREADS;
      const int nn = idx / (BK / 4), kk = (idx % (BK / 4)) * 4;
      if constexpr (BF16) *reinterpret_cast<bf16x4*>(B + nn * LDB + kk) = __builtin_convertvector(breg[j], bf16x4);
      else *reinterpret_cast<f32x4*>(B + nn * LDB + kk) = breg[j];
    }
  };

  for (int gi = 0; gi < L.ngroups; ++gi) {
    const PanelGroup& G = L.g[gi];
    const int w_rows = (G.n / G.seg) * G.seg_valid;
    const int w_bytes = (int)(((int64_t)(w_rows - 1) * G.ldw + G.k_len) * 4);
    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)G.w, 0, w_bytes, 0x00020000);
    const int kt0 = G.k_off / BK, kt1 = (G.k_off + G.k_len + BK - 1) / BK;
    for (int n0 = 0; n0 < G.n; n0 += BN) {
      f32x4 acc[FM][FN];
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      const bool wave_live = n0 + wn * 16 * FN < G.n;       // this wave's columns exist in the chunk (wave-uniform)
      load_b(G, wres, n0, kt0);
      __syncthreads();                                       // previous job's tiles (and, first time, the panel) are done
      store_b(0);
      if (kt0 + 1 < kt1) load_b(G, wres, n0, kt0 + 1);
      lds_barrier();
      for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (wave_live) {
          const elem_t* A = Ap + (size_t)(wm * 16 * FM + bi) * LDA + kt * BK;
          const elem_t* B = Bs + (size_t)buf * BN * LDB + (size_t)(wn * 16 * FN + bi) * LDB;
          if constexpr (BF16) {
#pragma unroll
            for (int ks = 0; ks < BK / 32; ++ks) {
              bf16x8 af[FM], bf[FN];
#pragma unroll
              for (int f = 0; f < FM; ++f) af[f] = *reinterpret_cast<const bf16x8*>(A + (size_t)f * 16 * LDA + ks * 32 + 8 * q);
#pragma unroll
              for (int f = 0; f < FN; ++f) bf[f] = *reinterpret_cast<const bf16x8*>(B + (size_t)f * 16 * LDB + ks * 32 + 8 * q);
#pragma unroll
              for (int fm = 0; fm < FM; ++fm)
#pragma unroll
                for (int fn = 0; fn < FN; ++fn)
                  acc[fm][fn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[fm], bf[fn], acc[fm][fn], 0, 0, 0);
            }
          } else {
#pragma unroll
            for (int ks = 0; ks < BK / 4; ++ks) {
              float af[FM], bf[FN];
#pragma unroll
              for (int f = 0; f < FM; ++f) af[f] = A[(size_t)f * 16 * LDA + ks * 4 + q];
#pragma unroll
              for (int f = 0; f < FN; ++f) bf[f] = B[(size_t)f * 16 * LDB + ks * 4 + q];
#pragma unroll
              for (int fm = 0; fm < FM; ++fm)
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = mma16x16x4(af[fm], bf[fn], acc[fm][fn]);
            }
          }
        }
        if (kt + 1 < kt1) {
          store_b(buf ^ 1);                                  // tile kt+1, in registers since the previous iteration
          if (kt + 2 < kt1) load_b(G, wres, n0, kt + 2);
        }
        lds_barrier();
      }
      // ---- epilogue of the job: bias, pad columns, plain stores
      if (wave_live) {
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
          const int col = n0 + wn * 16 * FN + fn * 16 + bi;
          if (col >= G.n) continue;
          const int sg = col / G.seg, u = col - sg * G.seg;
          const bool cv = u < G.seg_valid;
          float bsum = 0.0f;
          if (cv) {
            if (G.bias) bsum += G.bias[sg * G.seg_valid + u];
            if (G.bias2) bsum += G.bias2[sg * G.seg_valid + u];
          }
#pragma unroll
          for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = m0 + wm * 16 * FM + fm * 16 + q * 4 + r;
              if (row < L.M) {
                const float v = cv ? acc[fm][fn][r] + bsum : 0.0f;
                if (G.c_bf16) reinterpret_cast<__bf16*>(G.c)[(int64_t)row * G.ldc + col] = (__bf16)v;
                else G.c[(int64_t)row * G.ldc + col] = v;
              }
            }
        }
      }
    }
  }
  // zero-fill spans last (see gemm_common.h)
#pragma unroll
  for (int zi = 0; zi < MFM_GEMM_ZSPANS; ++zi) {
    if (L.zero_n[zi] > 0) {
      f32x4* z4 = reinterpret_cast<f32x4*>(L.zero_ptr[zi]);
      const int64_t n4 = L.zero_n[zi] >> 2;
      for (int64_t i = (int64_t)blockIdx.x * PANEL_THREADS + tid; i < n4; i += (int64_t)gridDim.x * PANEL_THREADS)
        z4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
}

// Panel height for a launch (0: none fits the LDS, or -- unless forced -- the tiled kernel is expected to be faster).
static int panel_height(const PanelLaunch& L, int precision, bool force, size_t* lds_out) {
  const bool bf16 = precision == 1;
  const int BK = bf16 ? 64 : 32;
  const int KT = cdiv(L.K, BK);
  const size_t esz = bf16 ? 2 : 4;
  const size_t lda = (size_t)KT * BK + (bf16 ? 8 : 4), ldb = (size_t)BK + (bf16 ? 8 : 4);
  // candidate heights (the instantiations below), cheapest first by rounds x (a + BM); a height whose panel does not
  // fit the 160 KB of LDS is skipped.  MFM_PANEL_BM forces one (tests, sweeps).
  static const int cand32[] = {64, 80}, cand16[] = {128, 160, 96};
  const int* cand = bf16 ? cand16 : cand32;
  const int ncand = bf16 ? 3 : 2;
  const double a = bf16 ? 128.0 : 59.0;      // fitted: fp32 rounds of 131 / 148 us at 64 / 80 rows, bf16 87 / 104 / 117 us at 96 / 128 / 160
  const int forced = opt_get("MFM_PANEL_BM") ? atoi(opt_get("MFM_PANEL_BM")) : 0;
  const long cus = device_cus();
  int BM = 0;
  size_t lds = 0;
  double best = 0.0;
  for (int i = 0; i < ncand; ++i) {
    const size_t need = ((size_t)cand[i] * lda + 2 * (size_t)PANEL_BN * ldb) * esz;
    if (need > 160 * 1024) continue;
    if (forced && cand[i] != forced) continue;
    const int panels = cdiv(L.M, cand[i]);
    const long rounds = (panels + cus - 1) / cus;
    const double cost = (double)rounds * (a + cand[i]);
    if (BM == 0 || cost < best) { BM = cand[i]; lds = need; best = cost; }
  }
  if (BM == 0) return 0;
  // against the tiled kernel, whose time grows with the rows alone (fp32 11.6 us, bf16 7.5 us per 1000 rows at the MOSI
  // sizes, where a panel row of one round costs 1.06 / 0.41 us): the panel kernel pays from rounds x (a + BM) <
  // 2.8 (fp32) / 4.7 (bf16) x rows per CU.  Both sides scale with the weight set, so the ratio carries to other shapes
  // (MOSEI, D = 409: only 64-row fp32 panels fit, and T*B = 20480 stays on the tiled kernel -- measured 1.35 vs 1.40 ms).
  if (!force && !forced && best >= (bf16 ? 4.7 : 2.8) * (double)L.M / (double)cus) return 0;
  *lds_out = lds;
  return BM;
}

bool gemm_panel_pays(const PanelLaunch& L, int precision, bool force) {
  size_t lds = 0;
  return L.M >= 1 && L.K >= 1 && panel_height(L, precision, force, &lds) > 0;
}

int gemm_panel_launch(PanelLaunch& L, const ZeroSpans* zs, int precision, bool force, hipStream_t stream) {
  MFM_REQUIRE(L.a && L.M >= 1 && L.K >= 1 && L.ngroups >= 1 && L.ngroups <= MFM_PANEL_MAXG, "gemm panel: bad launch (M=%d K=%d groups=%d)", L.M, L.K, L.ngroups);
  MFM_REQUIRE((int64_t)(L.M - 1) * L.lda + L.K < ((int64_t)1 << 29), "gemm panel: A spans >= 2^31 bytes");
  for (int i = 0; i < L.ngroups; ++i) {
    const PanelGroup& G = L.g[i];
    MFM_REQUIRE(G.w && G.c && G.n >= 1 && G.seg >= 1 && G.n % G.seg == 0 && G.seg_valid >= 1 && G.seg_valid <= G.seg && G.k_len >= 1 &&
                    G.k_off >= 0 && G.k_off + G.k_len <= L.K,
                "gemm panel: group %d: n=%d seg=%d/%d k=[%d,+%d) of %d", i, G.n, G.seg_valid, G.seg, G.k_off, G.k_len, L.K);
  }
  memset(L.zero_ptr, 0, sizeof(L.zero_ptr));
  memset(L.zero_n, 0, sizeof(L.zero_n));
  if (zs) {
    for (int i = 0; i < MFM_GEMM_ZSPANS; ++i) {
      if (zs->n[i] <= 0) continue;
      MFM_REQUIRE((zs->n[i] & 3) == 0 && (((uintptr_t)zs->ptr[i]) & 15) == 0, "gemm panel: zero span %d not 16-byte shaped", i);
      L.zero_ptr[i] = zs->ptr[i]; L.zero_n[i] = zs->n[i];
    }
  }
  const bool bf16 = precision == 1;
  for (int i = 0; i < L.ngroups; ++i) MFM_REQUIRE(bf16 || !L.g[i].c_bf16, "gemm panel: group %d: c_bf16 needs the bf16 kernel", i);
  size_t lds = 0;
  const int BM = panel_height(L, precision, force, &lds);
  if (BM == 0) { set_error("gemm panel: declined (K=%d does not fit the LDS, MFM_PANEL_BM names no built height, or the tiled kernel is cheaper)", L.K); return MFM_ERR_UNSUPPORTED; }
  const dim3 grid(cdiv(L.M, BM)), block(PANEL_THREADS);
#define MFM_PANEL_GO(B16, WM_, WN_, FM_, FN_)                                                                     \
  do {                                                                                                            \
    auto* fn = gemm_panel_kernel<B16, WM_, WN_, FM_, FN_>;                                                        \
    MFM_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));    \
    MFM_LAUNCH_TIMED(fn, grid, block, lds, stream, L);                                                          \
  } while (0)
  if (bf16) {
    if (BM == 128) MFM_PANEL_GO(true, 4, 2, 2, 4);
    else if (BM == 160) MFM_PANEL_GO(true, 2, 4, 5, 2);
    else MFM_PANEL_GO(true, 2, 4, 3, 2);
  } else {
    if (BM == 64) MFM_PANEL_GO(false, 2, 4, 2, 2);
    else MFM_PANEL_GO(false, 1, 8, 5, 1);
  }
#undef MFM_PANEL_GO
  MFM_LAUNCH_CHECK("gemm_panel_kernel");
  return MFM_OK;
}

}  // namespace mfm
