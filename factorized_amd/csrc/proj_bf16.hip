// Input projections of a bf16-RESIDENT plan at large batch (round 3): x W_ih^T + b_ih + b_hh for the four gates of every
// encoder LSTM, x fp32 in HBM, the gate pre-activations out as bf16 -- and, on the side, the padded bf16 image of the batch
// (x16) that the one-pass weight-gradient kernel (dw_bf16.hip) streams in the backward, so the separate x_to_bf16 launch
// and its second read of x disappear.
//
// Same decomposition as gemm_panel.hip (a workgroup keeps a panel of rows of x resident in LDS and walks every output
// column of every LSTM that consumes it), rebuilt around what bounded that kernel (profiles/r03_roofline_table_l_bf16.txt:
// 121 us for 147 MB of algorithmic traffic, 7.6 % MFMA utilisation): every workgroup streamed the fp32 weight set from
// L2 through registers (1.1 MB per workgroup against a per-CU ceiling of ~10 B/clk) and converted it to bf16 again, and
// wrote its bf16 results as 2-byte stores.  Here
//   * the weights are packed ONCE per step (proj_pack_kernel) into the bf16 tile image the workgroups consume: tiles of
//     [128 columns][32 k] in job order, gate padding, column ranges of the modality slices and the k tail resolved to
//     zeros, the 16-byte chunks of a tile placed so that the fragment reads are bank-conflict free -- so a tile arrives by
//     LDS-DMA (global_load_lds_dwordx4: no staging registers, no conversion), several tiles in flight, counted vmcnt and
//     a raw s_barrier per tile: 0.5 MB instead of 1.1 MB per workgroup;
//   * the tile requests live on a NINTH wave that does nothing else (see the kernel);
//   * the product is taken transposed (the weight fragment is the MFMA's A operand, the x fragment its B operand) and the
//     tile rows are permuted at pack time, so an accumulator lane holds 8 consecutive COLUMNS of one row: bias add (bias
//     image in LDS) and one 16-byte store per row fragment.
// Measured at B = 2048 (T*B = 40960 rows, MOSI sizes): 121 us (gemm_panel) + 28 us (x_to_bf16) -> 57 us.  What is left is the
// per-CU traffic itself: 208 KB of x + 488 KB of tiles in, 307 KB of gates + 110 KB of x16 out = 1.1 MB per workgroup at
// the ~10 B/clk a CU exchanges with the L2 (49 us); the tile stream alone (MFM_PROJ16_DBG=15) takes 22 us whatever the
// pipeline depth (3, 4, 5 stages measured equal).
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "internal.h"
#include "pack_dev.h"

namespace mfm {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int PJ_THREADS = 512;
constexpr int PJ_MAXG = MFM_PANEL_MAXG;

struct PjGroupDev { __bf16* c; int64_t ldc; int n, kt0, kt1, bias_off; };
struct PjDev {
  const float* x; int64_t lda; int M, K, KP;
  const __bf16* wimg; const float* bimg;
  __bf16* x16; int x16_ld; int xsrc0[3], xn[3], xdst0[3];
  PjGroupDev g[PJ_MAXG]; int ngroups, ntiles, nbias, S, dbg;
  int ns, tb[9];                 // column splits (few row panels): workgroup (panel, y) runs the jobs whose tiles are [tb[y], tb[y + 1])
  float* zero_ptr[MFM_GEMM_ZSPANS]; int64_t zero_n[MFM_GEMM_ZSPANS];
};
__global__ __launch_bounds__(256) void proj_pack_kernel(const PjPackDev L) {
  proj_pack_body(L, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

// every weight image of the step in one launch: blocks [0, b1) the recurrences' fragments, [b1, b2) the projection tiles
// and biases, [b2, ..) the decoders' fc1 images (pack_dev.h)
struct PackAllDev { PackLaunch lstm; PjPackDev proj; Fc1PackArgs fc1; int b1, b2; };
__global__ __launch_bounds__(256) void pack_all_kernel(const PackAllDev A) {
  const int b = (int)blockIdx.x;
  if (b < A.b1) lstm_pack_body(A.lstm, (int64_t)b * 256 + threadIdx.x);
  else if (b < A.b2) proj_pack_body(A.proj, (int64_t)(b - A.b1) * 256 + threadIdx.x);
  else fc1_pack_body(A.fc1, (int64_t)(b - A.b2) * 256 + threadIdx.x);
}

// 8 compute waves = 2 (rows) x 4 (columns), a wave owns FM x 2 fragments of 16 x 16: BM = 32 FM rows, 128 columns per job;
// a NINTH wave only requests weight tiles (LDS-DMA) and waits for them.  Loads and stores share vmcnt on gfx9 and complete
// out of order with respect to each other, so a wave that both waits for its tiles by count and stores results has to
// drain its stores at every wait (measured: 15 of 72 us with 16 jobs per workgroup); with the requests on a wave of their
// own the compute waves never wait on vector memory inside the loop and their stores retire in the background.
template <int FM>
__global__ __launch_bounds__(PJ_THREADS + 64) void proj_bf16_kernel(const PjDev L) {
  constexpr int FN = 2, BM = 32 * FM;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int KP = L.KP, LDA = KP + 8;
  __bf16* Ap = reinterpret_cast<__bf16*>(smem);                                        // [BM][LDA]
  const unsigned a_bytes_lds = (unsigned)(((size_t)BM * LDA * 2 + 1023) / 1024 * 1024);
  unsigned char* Bt = smem + a_bytes_lds;                                              // [S][8192]
  const int S = L.S;
  float* Bias = reinterpret_cast<float*>(Bt + (size_t)S * (PJ_TILE * 2));             // [nbias]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // (wave-uniform, and the compiler must know: M0 is scalar)
  // few row panels (small T*B): `ns` workgroups share a panel, each with a contiguous range of the jobs -- it streams only that
  // range's weight tiles (a workgroup that walks all 61 tiles of the MOSI plan takes ~22 us for them whatever its panel height)
  const int ns = L.ns, panel = (int)blockIdx.x / ns, ysplit = (int)blockIdx.x - panel * ns;
  const int tb = L.tb[ysplit], te = L.tb[ysplit + 1];
  const int m0 = panel * BM;

  if (wave == 8) {
    // ---- the requesting wave.  A tile is 8 instructions of 64 lanes x 16 bytes; the LDS address of a lane is
    // M0 + 16 x lane.  Tiles past the end re-request the last tile into a slot nobody reads any more, so that every
    // iteration issues the same number of instructions and the counted wait stays valid.
    auto issue = [&](int t) {
      const int tt = min(t, te - 1);
      const __bf16* g = L.wimg + (int64_t)tt * PJ_TILE + lane * 8;
      const unsigned base = (unsigned)(uintptr_t)(lds_void*)(Bt + ((t - tb) % S) * (PJ_TILE * 2));
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const __bf16* gj = g + j * 512;
        const unsigned ldsaddr = base + j * 1024;
        { unsigned m0_saved;    // M0 is the compiler's to manage (clobbering a reserved register is undefined behaviour): saved and restored here
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                       : "=&s"(m0_saved) : "v"(gj), "s"(ldsaddr) : "memory"); }
      }
    };
    for (int t = tb; t < tb + S - 1; ++t) issue(t);
    asm volatile("s_barrier" ::: "memory");                  // (the compute waves' barrier after the panel)
    for (int t = tb; t < te; ++t) {
      // tile t is the oldest outstanding one: wait until only the 8 (S - 2) instructions of the younger tiles remain
      switch (S) {
        case 6: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
      asm volatile("s_barrier" ::: "memory");                // tile t is in LDS; the compute waves are done with tile t-1
      issue(t + S - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the trailing requests must land before the LDS is released
    return;
  }

  const int bi = lane & 15, q = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;
  // ---- the panel: rows m0 .. m0+BM-1, columns 0 .. KP-1 of x (zero beyond K / M), fp32 -> bf16
  {
    const int x_bytes = (int)(((int64_t)(L.M - 1) * L.lda + L.K) * 4);
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)L.x, 0, x_bytes, 0x00020000);
    const int gpr = KP / 4;
    const int total = BM * gpr;
    // idx / gpr by a multiply (exact while idx x gpr < 2^24; the product stays below 2^32 for BM <= 255): a run-time integer
    // division is ~40 VALU instructions, and this loop had two per 16-byte piece -- ~56 per thread ahead of the first MFMA
    const unsigned gmagic = (1u << 24) / (unsigned)gpr + 1u;
    auto div_gpr = [&](int idx) { return (int)(((unsigned)idx * gmagic) >> 24); };
    constexpr int U = 14;                         // 16-byte loads in flight per thread (HBM latency x 64 B/clk wants ~100 KB)
    for (int base = tid; base < ((L.dbg & 8) ? 0 : total); base += PJ_THREADS * U) {
      f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = min(base + u * PJ_THREADS, total - 1);
        const int r = div_gpr(idx), k = (idx - r * gpr) * 4;
        const int off = (int)(((int64_t)min(m0 + r, L.M - 1) * L.lda + min(k, L.K - 1)) * 4);
        v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, off, 0, 0));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = base + u * PJ_THREADS;
        if (idx < total) {
          const int r = div_gpr(idx), k = (idx - r * gpr) * 4;
          f32x4 w = v[u];
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = (m0 + r < L.M && k + e < L.K) ? w[e] : 0.0f;
          *reinterpret_cast<bf16x4*>(Ap + (size_t)r * LDA + k) = __builtin_convertvector(w, bf16x4);
        }
      }
    }
    for (int i = tid; i < L.nbias; i += PJ_THREADS) Bias[i] = L.bimg[i];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // ---- side output: the padded bf16 image of these rows (modality slices on 16-column boundaries, pads zero)
  if (L.x16 && ysplit == 0 && !(L.dbg & 1)) {
    const int cpr = L.x16_ld >> 3;
    const unsigned cmagic = (1u << 24) / (unsigned)cpr + 1u;
    for (int idx = tid; idx < BM * cpr; idx += PJ_THREADS) {
      const int r = (int)(((unsigned)idx * cmagic) >> 24), c8 = (idx - r * cpr) * 8;
      if (m0 + r >= L.M) continue;
      int s = 0;
      if (c8 >= L.xdst0[1]) s = 1;
      if (c8 >= L.xdst0[2]) s = 2;
      const int rel = c8 - L.xdst0[s], src = L.xsrc0[s] + rel, nv = L.xn[s] - rel;      // nv: valid elements from here on
      const __bf16* p = Ap + (size_t)r * LDA;
      bf16x8 v;
      if (nv >= 8 && (src & 7) == 0) {
        v = *reinterpret_cast<const bf16x8*>(p + src);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (e < nv) ? p[min(src + e, KP - 1)] : (__bf16)0.0f;
      }
      *reinterpret_cast<bf16x8*>(L.x16 + (int64_t)(m0 + r) * L.x16_ld + c8) = v;
    }
  }

  // ---- jobs: (group, 128-column chunk) x the k tiles of the group's column range; the image holds them in this order
  int t = 0;
  for (int gi = 0; gi < L.ngroups; ++gi) {
    const PjGroupDev G = L.g[gi];
    for (int n0 = 0; n0 < G.n; n0 += PJ_BN) {
      if (t < tb || t >= te) { t += G.kt1 - G.kt0; continue; }      // another split's job
      const int colw = n0 + wn * 32;                       // this wave's first column of the chunk
      const bool wave_live = colw < G.n;
      f32x4 acc[FM][FN];
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int kt = G.kt0; kt < G.kt1; ++kt, ++t) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // tile t landed (the requesting wave waited for it)
        if (wave_live && !(L.dbg & 4)) {
          const unsigned char* B = Bt + ((t - tb) % S) * (PJ_TILE * 2);
          bf16x8 wf[FN], xf[FM];
#pragma unroll
          for (int fn = 0; fn < FN; ++fn) {
            const int nn = wn * 32 + fn * 16 + bi;
            wf[fn] = *reinterpret_cast<const bf16x8*>(B + pj_tile_slot(nn, q) * 16);
          }
#pragma unroll
          for (int fm = 0; fm < FM; ++fm)
            xf[fm] = *reinterpret_cast<const bf16x8*>(Ap + (size_t)((wm * FM + fm) * 16 + bi) * LDA + kt * PJ_BK + 8 * q);
#pragma unroll
          for (int fm = 0; fm < FM; ++fm)
#pragma unroll
            for (int fn = 0; fn < FN; ++fn)
              acc[fm][fn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[fn], xf[fm], acc[fm][fn], 0, 0, 0);
        }
      }
      // the job's results.  Tile row 16 fn + i of this wave's 32 carries column 8 (i / 4) + 4 fn + i % 4 (proj_pack_kernel),
      // and accumulator register r of lane (bi, q) is [MFMA row 4q + r][x row bi]: the lane holds columns 8q .. 8q+7 of
      // row bi -- bias add, one 16-byte store per row fragment
      if (wave_live && !(L.dbg & 2)) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(Bias + G.bias_off + colw + 8 * q);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(Bias + G.bias_off + colw + 8 * q + 4);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
          const int row = m0 + (wm * FM + fm) * 16 + bi;
          const bf16x4 lo = __builtin_convertvector(acc[fm][0] + b0, bf16x4), hi = __builtin_convertvector(acc[fm][1] + b1, bf16x4);
          if (row < L.M) *reinterpret_cast<bf16x8*>(G.c + (int64_t)row * G.ldc + colw + 8 * q) = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      }
    }
  }
#pragma unroll
  for (int zi = 0; zi < MFM_GEMM_ZSPANS; ++zi) {
    if (L.zero_n[zi] > 0) {
      f32x4* z4 = reinterpret_cast<f32x4*>(L.zero_ptr[zi]);
      const int64_t n4 = L.zero_n[zi] >> 2;
      for (int64_t i = (int64_t)blockIdx.x * PJ_THREADS + tid; i < n4; i += (int64_t)gridDim.x * PJ_THREADS)
        z4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
}

}  // namespace

int proj_bf16_plan(const PanelLaunch& L, ProjPlan* out) {
  memset(out, 0, sizeof(*out));
  if (L.ngroups < 1 || L.ngroups > PJ_MAXG || L.K < 1) return 0;
  int t = 0, nb = 0;
  for (int i = 0; i < L.ngroups; ++i) {
    const PanelGroup& G = L.g[i];
    if (G.n < 1 || G.seg < 1 || G.n % G.seg || (G.n & 63) || G.seg_valid < 1 || G.seg_valid > G.seg || G.k_len < 1 || G.k_off < 0 ||
        G.k_off + G.k_len > L.K)
      return 0;
    out->kt0[i] = G.k_off / PJ_BK;
    out->nkt[i] = cdiv(G.k_off + G.k_len, PJ_BK) - out->kt0[i];
    out->nchunks[i] = cdiv(G.n, PJ_BN);
    out->tile0[i] = t; out->bias_off[i] = nb;
    t += out->nchunks[i] * out->nkt[i];
    nb += out->nchunks[i] * PJ_BN;
  }
  out->ntiles = t; out->nbias = nb;
  // panel height and pipeline depth that fit the LDS: rounds x (a + BM), a = the per-workgroup cost of the weight stream
  const int KP = round_up(L.K, PJ_BK);
  static const int cand[] = {160, 128, 96, 64};
  const long cus = device_cus();
  const int forced = opt_get("MFM_PROJ16_BM") ? atoi(opt_get("MFM_PROJ16_BM")) : 0;
  double best = 0.0;
  for (int i = 0; i < 4; ++i) {
    if (forced && cand[i] != forced) continue;
    const size_t a_lds = ((size_t)cand[i] * (KP + 8) * 2 + 1023) / 1024 * 1024;
    int S = 6;
    if (const char* e = opt_get("MFM_PROJ16_STAGES")) S = std::max(3, std::min(6, atoi(e)));
    while (S >= 3 && a_lds + (size_t)S * PJ_TILE * 2 + (size_t)nb * 4 > 160 * 1024) --S;
    if (S < 3) continue;
    // column splits (round 5): with few row panels, ns workgroups share a panel and stream 1 / ns of the tiles each
    const long npanels = cdiv(std::max(L.M, 1), cand[i]);
    int jobs = 0;
    for (int g = 0; g < L.ngroups; ++g) jobs += out->nchunks[g];
    int ns = (int)std::max<long>(1, std::min<long>(std::min(8, jobs), cus / npanels));
    if (const char* e = opt_get("MFM_PROJ16_NS")) ns = std::max(1, std::min(std::min(8, jobs), atoi(e)));
    const long rounds = (npanels * ns + cus - 1) / cus;
    const double cost = (double)rounds * (100.0 / ns + cand[i]);
    if (out->BM == 0 || cost < best) { out->BM = cand[i]; out->S = S; out->ns = ns; out->lds = a_lds + (size_t)S * PJ_TILE * 2 + (size_t)nb * 4; best = cost; }
  }
  return out->BM != 0;
}

int proj_pack_prepare(const PanelLaunch& L, const ProjPlan& P, void* wimg, float* bimg, PjPackDev* out) {
  MFM_REQUIRE(wimg && bimg && (((uintptr_t)wimg) & 15) == 0 && (((uintptr_t)bimg) & 15) == 0, "proj bf16 pack: bad scratch");
  PjPackDev& D = *out;
  memset(&D, 0, sizeof(D));
  for (int i = 0; i < L.ngroups; ++i) {
    const PanelGroup& G = L.g[i];
    PjPackGroup& d = D.g[i];
    d.w = G.w; d.bias = G.bias; d.bias2 = G.bias2; d.ldw = G.ldw;
    d.n = G.n; d.seg = G.seg; d.seg_valid = G.seg_valid; d.k_off = G.k_off; d.k_len = G.k_len;
    d.kt0 = P.kt0[i]; d.nkt = P.nkt[i]; d.tile0 = P.tile0[i]; d.nchunks = P.nchunks[i]; d.bias_off = P.bias_off[i];
  }
  D.ngroups = L.ngroups; D.ntiles = P.ntiles; D.nbias = P.nbias;
  D.wimg = reinterpret_cast<__bf16*>(wimg); D.bimg = bimg;
  return MFM_OK;
}

int proj_bf16_pack_launch(const PanelLaunch& L, const ProjPlan& P, void* wimg, float* bimg, hipStream_t stream) {
  PjPackDev D;
  const int rc = proj_pack_prepare(L, P, wimg, bimg, &D);
  if (rc != MFM_OK) return rc;
  const int64_t total = (int64_t)P.ntiles * (PJ_TILE / 8) + P.nbias;
  MFM_LAUNCH_TIMED(proj_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, D);
  MFM_LAUNCH_CHECK("proj_pack_kernel");
  return MFM_OK;
}

// any of the three parts may be null / empty
int pack_all_launch(const PackLaunch* lstm, const PjPackDev* proj, const Fc1PackArgs* fc1, hipStream_t stream) {
  PackAllDev A;
  memset(&A, 0, sizeof(A));
  int64_t n1 = 0, n2 = 0, n3 = 0;
  if (lstm && lstm->count > 0) { A.lstm = *lstm; n1 = (lstm->total + 255) / 256; }
  if (proj && proj->ntiles > 0) { A.proj = *proj; n2 = ((int64_t)proj->ntiles * (PJ_TILE / 8) + proj->nbias + 255) / 256; }
  if (fc1 && fc1->n > 0) { A.fc1 = *fc1; n3 = (fc1->begin[fc1->n] + 255) / 256; }
  if (n1 + n2 + n3 == 0) return MFM_OK;
  MFM_REQUIRE(n1 + n2 + n3 < ((int64_t)1 << 30), "pack: too many blocks");
  A.b1 = (int)n1; A.b2 = (int)(n1 + n2);
  MFM_LAUNCH_TIMED(pack_all_kernel, dim3((unsigned)(n1 + n2 + n3)), dim3(256), 0, stream, A);
  MFM_LAUNCH_CHECK("pack_all_kernel");
  return MFM_OK;
}

int proj_bf16_launch(const PanelLaunch& L, const ProjPlan& P, const void* wimg, const float* bimg, void* x16, int x16_ld,
                     const int* xsrc0, const int* xn, const int* xdst0, const ZeroSpans* zs, hipStream_t stream) {
  MFM_REQUIRE(L.a && L.M >= 1 && P.BM > 0 && P.ntiles >= 1, "proj bf16: bad launch");
  MFM_REQUIRE((int64_t)(L.M - 1) * L.lda + L.K < ((int64_t)1 << 29), "proj bf16: x spans >= 2^31 bytes");
  PjDev D;
  memset(&D, 0, sizeof(D));
  D.x = L.a; D.lda = L.lda; D.M = L.M; D.K = L.K; D.KP = round_up(L.K, PJ_BK);
  D.wimg = reinterpret_cast<const __bf16*>(wimg); D.bimg = bimg;
  D.x16 = reinterpret_cast<__bf16*>(x16); D.x16_ld = x16_ld;
  if (x16) {
    MFM_REQUIRE((x16_ld & 7) == 0 && (((uintptr_t)x16) & 15) == 0, "proj bf16: x16 image not 16-byte shaped");
    int end = 0;
    for (int s = 0; s < 3; ++s) {
      MFM_REQUIRE(xdst0[s] == end && xn[s] >= 1 && xsrc0[s] >= 0 && xsrc0[s] + xn[s] <= L.K, "proj bf16: x16 slice %d", s);
      end += round_up(xn[s], 16);
      D.xsrc0[s] = xsrc0[s]; D.xn[s] = xn[s]; D.xdst0[s] = xdst0[s];
    }
    MFM_REQUIRE(end == x16_ld, "proj bf16: x16 slices do not cover the image (%d of %d columns)", end, x16_ld);
  }
  for (int i = 0; i < L.ngroups; ++i) {
    const PanelGroup& G = L.g[i];
    MFM_REQUIRE(G.c && G.c_bf16 && (G.ldc & 7) == 0 && (G.n & 63) == 0 && (((uintptr_t)G.c) & 15) == 0, "proj bf16: group %d: output must be a bf16 buffer with 16-byte rows, 64-column gates", i);
    PjGroupDev& d = D.g[i];
    d.c = reinterpret_cast<__bf16*>(G.c); d.ldc = G.ldc; d.n = G.n; d.kt0 = P.kt0[i]; d.kt1 = P.kt0[i] + P.nkt[i]; d.bias_off = P.bias_off[i];
  }
  D.ngroups = L.ngroups; D.ntiles = P.ntiles; D.nbias = P.nbias; D.S = P.S;
  D.dbg = opt_get("MFM_PROJ16_DBG") ? atoi(opt_get("MFM_PROJ16_DBG")) : 0;     // tuning aid: skip parts of the kernel
  if (zs) {
    for (int i = 0; i < MFM_GEMM_ZSPANS; ++i) {
      if (zs->n[i] <= 0) continue;
      MFM_REQUIRE((zs->n[i] & 3) == 0 && (((uintptr_t)zs->ptr[i]) & 15) == 0, "proj bf16: zero span %d not 16-byte shaped", i);
      D.zero_ptr[i] = zs->ptr[i]; D.zero_n[i] = zs->n[i];
    }
  }
  // column splits: the jobs (group, 128-column chunk) are dealt to `ns` workgroups per panel in contiguous ranges of about equal
  // tile counts (MFM_PROJ16_NS forces a count when the plan is built, 1 = off)
  const int npanels = cdiv(L.M, P.BM);
  {
    int jobs = 0;
    for (int i = 0; i < L.ngroups; ++i) jobs += P.nchunks[i];
    const int ns = std::max(1, std::min(P.ns, std::min(jobs, 8)));           // chosen with the panel height (proj_bf16_plan)
    D.ns = ns; D.tb[0] = 0;
    int y = 1, t = 0;
    for (int i = 0; i < L.ngroups && y < ns; ++i)
      for (int c = 0; c < P.nchunks[i] && y < ns; ++c) {
        t += P.nkt[i];
        // close split y - 1 behind this job when it has reached its share (and enough jobs remain for the other splits)
        int left = 0;
        for (int i2 = i; i2 < L.ngroups; ++i2) left += (i2 == i) ? P.nchunks[i2] - c - 1 : P.nchunks[i2];
        if (t * ns >= y * P.ntiles || left <= ns - y) { D.tb[y++] = t; }
      }
    for (; y <= ns; ++y) D.tb[y] = P.ntiles;
    for (int k = 0; k < ns; ++k) MFM_REQUIRE(D.tb[k] < D.tb[k + 1], "proj bf16: empty column split %d of %d", k, ns);
  }
  const dim3 grid(npanels * D.ns), block(PJ_THREADS + 64);
#define MFM_PJ_GO(FM_)                                                                                              \
  do {                                                                                                              \
    auto* fn = proj_bf16_kernel<FM_>;                                                                               \
    MFM_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.lds));    \
    MFM_LAUNCH_TIMED(fn, grid, block, P.lds, stream, D);                                                          \
  } while (0)
  switch (P.BM) {
    case 160: MFM_PJ_GO(5); break;
    case 128: MFM_PJ_GO(4); break;
    case 96: MFM_PJ_GO(3); break;
    default: MFM_PJ_GO(2); break;
  }
#undef MFM_PJ_GO
  MFM_LAUNCH_CHECK("proj_bf16_kernel");
  return MFM_OK;
}

}  // namespace mfm
