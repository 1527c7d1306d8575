// Grouped, batched, strided fp32 GEMM on v_mfma_f32_16x16x4_f32 (gfx950).
//
// One launch covers up to MFM_GEMM_MAXP independent problems (input projections of the 4
// encoders; the 12 weight-gradient products of the encoder backward; ...), because at the
// MOSI batch size every one of them is far too small to fill 256 CUs on its own and a launch
// boundary costs ~1.5 us.  Tiles are 32x32 or 64x64 (picked so the group yields >= ~2 blocks
// per CU), K is staged 32 deep through LDS in [k][m] / [k][n] order so that the MFMA operand
// reads (16 consecutive m at one k) are bank-conflict free (row stride = tile+16 dwords, i.e.
// == 16 mod 32 banks for the two k rows a 32-lane group touches).
#include <stdlib.h>

#include "common.h"

namespace mfm {

#define MFM_GEMM_MAXP 24   // 24 x 168 B of descriptors stay inside the 4 KB kernel-argument segment
constexpr int BK = 32;   // K depth of one LDS stage: 8 MFMA k-steps between barriers

struct GemmProblem {
  MfmGemmDesc d;
  int tiles_m, tiles_n, block_begin, k_per_split;
};
struct GemmGroup {
  GemmProblem p[MFM_GEMM_MAXP];
  int count;
};

template <int FR>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmGroup g) {
  constexpr int BM = 32 * FR, BN = 32 * FR;
  constexpr int LDA = BM + 16, LDB = BN + 16;
  constexpr int EPT_A = BM * BK / 256, EPT_B = BN * BK / 256;
  __shared__ float As[BK * LDA];
  __shared__ float Bs[BK * LDB];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int bi = lane & 15, q = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- locate problem / tile (wave-uniform)
  int pi = 0;
  const int bid = blockIdx.x;
#pragma unroll 1
  for (int i = 1; i < g.count; ++i)
    if (bid >= g.p[i].block_begin) pi = i;
  const GemmProblem& P = g.p[pi];
  const MfmGemmDesc& d = P.d;
  int local = bid - P.block_begin;
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

  // Register ring of DEPTH K-tiles in flight: at these sizes (K = 300..640, a handful of MFMAs per
  // tile) the K loop is a chain of global-load round trips, so the loads of tile kt+DEPTH are
  // issued as soon as tile kt has been copied to LDS.
  constexpr int DEPTH = 3;
  float ra[DEPTH][EPT_A], rb[DEPTH][EPT_B];

  auto load_tiles = [&](int slot, int k0) {
#pragma unroll
    for (int j = 0; j < EPT_A; ++j) {
      const int idx = tid * EPT_A + j;
      int m, k;
      if (a_mcontig) { k = idx / BM; m = idx % BM; } else { m = idx / BK; k = idx % BK; }
      const int gm = m0 + m, gk = k0 + k;
      // unconditional load from a clamped (always valid) address + select: no branch, so the
      // compiler can keep several tiles' loads in flight with counted vmcnt waits
      const float v = A[(int64_t)min(gm, d.m - 1) * d.a_sm + (int64_t)min(gk, kend - 1) * d.a_sk];
      ra[slot][j] = v * (float)((int)(gm < d.m) & (int)(gk < kend));   // multiply, not select: stays branch-free
    }
#pragma unroll
    for (int j = 0; j < EPT_B; ++j) {
      const int idx = tid * EPT_B + j;
      int n, k;
      if (b_ncontig) { k = idx / BN; n = idx % BN; } else { n = idx / BK; k = idx % BK; }
      const int gn = n0 + n, gk = k0 + k;
      const float v = Bm[(int64_t)min(gk, kend - 1) * d.b_sk + (int64_t)min(gn, d.n_valid - 1) * d.b_sn];
      rb[slot][j] = v * (float)((int)(gn < d.n_valid) & (int)(gk < kend));
    }
  };
  auto store_tiles = [&](int slot) {
#pragma unroll
    for (int j = 0; j < EPT_A; ++j) {
      const int idx = tid * EPT_A + j;
      int m, k;
      if (a_mcontig) { k = idx / BM; m = idx % BM; } else { m = idx / BK; k = idx % BK; }
      As[k * LDA + m] = ra[slot][j];
    }
#pragma unroll
    for (int j = 0; j < EPT_B; ++j) {
      const int idx = tid * EPT_B + j;
      int n, k;
      if (b_ncontig) { k = idx / BN; n = idx % BN; } else { n = idx / BK; k = idx % BK; }
      Bs[k * LDB + n] = rb[slot][j];
    }
  };

  f32x4 acc[FR][FR];
#pragma unroll
  for (int i = 0; i < FR; ++i)
#pragma unroll
    for (int j = 0; j < FR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt = (kend - kbeg + BK - 1) / BK;
#pragma unroll
  for (int s = 0; s < DEPTH; ++s)
    if (s < nkt) load_tiles(s, kbeg + s * BK);
  for (int kt0 = 0; kt0 < nkt; kt0 += DEPTH) {
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
      const int kt = kt0 + s;
      if (kt < nkt) {
        store_tiles(s);
        __syncthreads();
        if (kt + DEPTH < nkt) load_tiles(s, kbeg + (kt + DEPTH) * BK);
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
          float af[FR], bf[FR];
#pragma unroll
          for (int f = 0; f < FR; ++f) {
            af[f] = As[(ks * 4 + q) * LDA + wm * 16 * FR + f * 16 + bi];
            bf[f] = Bs[(ks * 4 + q) * LDB + wn * 16 * FR + f * 16 + bi];
          }
#pragma unroll
          for (int fm = 0; fm < FR; ++fm)
#pragma unroll
            for (int fn = 0; fn < FR; ++fn) acc[fm][fn] = mma16x16x4(af[fm], bf[fn], acc[fm][fn]);
        }
        __syncthreads();
      }
    }
  }

  // ---- epilogue
  float* __restrict__ C = d.c + (int64_t)z * d.c_sz;
  float* __restrict__ C2 = d.c2 ? d.c2 + (int64_t)z * d.c_sz : nullptr;
#pragma unroll
  for (int fm = 0; fm < FR; ++fm)
#pragma unroll
    for (int fn = 0; fn < FR; ++fn) {
      const int col = n0 + wn * 16 * FR + fn * 16 + bi;
      if (col >= d.n) continue;
      float bsum = 0.0f;
      if (split == 0 && col < d.n_valid) {
        if (d.bias) bsum += d.bias[(int64_t)z * d.bias_sz + col];
        if (d.bias2) bsum += d.bias2[(int64_t)z * d.bias_sz + col];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 16 * FR + fm * 16 + q * 4 + r;
        if (row >= d.m) continue;
        float v = (col < d.n_valid) ? d.alpha * acc[fm][fn][r] + bsum : 0.0f;
        const int64_t off = (int64_t)row * d.ldc + col;
        if (d.accumulate) {
          if (col < d.n_valid) {
            atomicAdd(C + off, v);
            if (C2) atomicAdd(C2 + off, v);
          }
        } else {
          C[off] = v;
          if (C2) C2[off] = v;
        }
      }
    }
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
int gemm_group_launch(const MfmGemmDesc* descs, int count, hipStream_t stream) {
  MFM_REQUIRE(count >= 1 && count <= MFM_GEMM_MAXP, "gemm group: count %d out of range", count);
  const int cus = device_cus();
  GemmGroup g;
  memset(&g, 0, sizeof(g));
  g.count = count;
  // pass 1: block count with 64x64 tiles and no split; decide tile size
  long blocks64 = 0;
  for (int i = 0; i < count; ++i) {
    const MfmGemmDesc& d = descs[i];
    MFM_REQUIRE(d.m > 0 && d.n > 0 && d.k >= 0 && d.batch > 0, "gemm[%d]: bad dims m=%d n=%d k=%d batch=%d", i, d.m, d.n, d.k, d.batch);
    MFM_REQUIRE(d.a && d.b && d.c, "gemm[%d]: null operand", i);
    MFM_REQUIRE(d.n_valid >= 0 && d.n_valid <= d.n, "gemm[%d]: n_valid %d > n %d", i, d.n_valid, d.n);
    MFM_REQUIRE(d.split_k <= 1 || d.accumulate, "gemm[%d]: split_k needs accumulate", i);
    blocks64 += (long)cdiv(d.m, 64) * cdiv(d.n, 64) * d.batch;
  }
  int FR = (blocks64 >= 2L * cus) ? 2 : 1;
  if (const char* e = getenv("MFM_GEMM_FR")) FR = (e[0] == '2') ? 2 : 1;   // tuning override
  const int BT = 32 * FR;
  long base_blocks = 0;
  for (int i = 0; i < count; ++i)
    base_blocks += (long)cdiv(descs[i].m, BT) * cdiv(descs[i].n, BT) * descs[i].batch;
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
        const long want_depth = (P.d.k + 1023) / 1024;
        long want = want_fill > want_depth ? want_fill : want_depth;
        const long maxs = P.d.k / 64;
        split = (int)(want < 1 ? 1 : (want > maxs ? maxs : want));
        if (split < 1) split = 1;
      }
    }
    int kps = round_up(cdiv(P.d.k > 0 ? P.d.k : 1, split), BK);
    split = cdiv(P.d.k > 0 ? P.d.k : 1, kps);
    P.d.split_k = split;
    P.k_per_split = kps;
    P.block_begin = total;
    total += P.tiles_m * P.tiles_n * P.d.batch * split;
  }
  if (FR == 2)
    hipLaunchKernelGGL(gemm_f32_kernel<2>, dim3(total), dim3(256), 0, stream, g);
  else
    hipLaunchKernelGGL(gemm_f32_kernel<1>, dim3(total), dim3(256), 0, stream, g);
  MFM_LAUNCH_CHECK("gemm_f32_kernel");
  return MFM_OK;
}

}  // namespace mfm

extern "C" int mfm_gemm_grouped_f32(const MfmGemmDesc* descs, int count, void* stream) {
  if (!descs || count <= 0) {
    mfm::set_error("mfm_gemm_grouped_f32: no problems");
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
