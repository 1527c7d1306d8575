// Decoder output layer, forward AND backward-to-hidden in one launch (fp32 plans).
//
// Reference: decoderLSTM.fc1 applied to every decoder hidden state (mfm_model.py:40-63), the three reconstruction
// losses `lda_x* * mse(x_hat, x)` (mfm_mosi.py:441-446) and, in training, what autograd sends back through fc1 to the
// hidden states.  In the plan these were two grouped-GEMM launches on the dependency chain (F4 with the squared-error
// epilogue, B0); at the reference's B=32 each is a ~10 us launch of which most is latency.  Nothing else separates
// them: d x_hat is an elementwise function of x_hat, so a workgroup that owns 16 (t, b) rows of one decoder can run
//
//   x_hat = H Wfc^T + b  ->  diff = x_hat - x  ->  loss += sum diff^2 / count,  dx_hat = 2 lda / count * diff
//   dH    = dx_hat Wfc                                                          (training only)
//
// back to back with dx_hat staged in LDS.  Wfc is read from L2 twice (once per product), never staged: a workgroup
// uses every element exactly once per product.  The weight-gradient products dWfc = dx_hat^T H and dbfc stay in the
// step's tail GEMM launch (they only feed the optimizer).
//
// Decomposition: one workgroup = 16 rows x one group of <= 8 output fragments (16 columns each) of one decoder.  Decoder l
// (300 columns, 19 fragments) takes 3 column groups per row tile; their contributions to dH are partial sums over the
// columns and are ADDED to dH with atomics -- the caller puts the dH block into the step's zero spans.  (A single column group
// stores instead.)
//
// 512 threads = 8 waves; MFMA 16x16x4 fp32 tiles.
//   product 1: wave w owns output fragment w of the group; the reduction over hidden units walks 16-wide blocks: lane
//              (bi, q) takes the four units 16 j + 4 q + {0..3} of its row of H (LDS, one 16-byte read) and of row
//              n = bi of Wfc (one 16-byte buffer load) and feeds four MFMAs -- the order of the reduction index inside a
//              block is free as long as both operands agree.
//   product 2: the reduction over the group's columns is split into 8 contiguous ranges, one per wave, every wave
//              accumulating all Hp/16 output fragments; the 8 partial tiles are summed through LDS in a fixed order
//              (the sums over column GROUPS are atomics, so dH is reproducible to rounding order only when spread).
//   Every global operand of both products (weights, targets, bias) is requested before the first MFMA: the workgroup is
//   alone on its CU and a dependent load per reduction step would cost an L2 round trip each.
#include <hip/hip_runtime.h>
#include "internal.h"
#include "lstamp.h"

namespace mfm {

constexpr int FC1_THREADS = 512;
constexpr int FC1_WAVES = 8;
constexpr int FC1_ROWS = 16;
constexpr int FC1_MAXF = 8;                  // Hp <= 128; <= 8 output fragments per column group
constexpr int FC1_KS2W = 4;                  // reduction steps of product 2 per wave: 4 * 8 fragments / 8 waves
constexpr int FC1_LD = 16 * FC1_MAXF + 4;    // LDS row stride: (LD / 4) odd and LD == 4 (mod 64): conflict-free reads
constexpr int FC1_OOB = 0x7FFFFFF0;          // buffer offset beyond any resource: the load returns 0

// bf16 plans: the same products with every operand (H, Wfc, dx_hat) rounded to bf16 first and fp32 accumulation -- what
// the bf16-operand GEMM computes; at these sizes the launch is latency, not matrix time, so the fp32 MFMA stays
__device__ __forceinline__ float rnd_bf16(float x, bool on) { return on ? (float)(__bf16)x : x; }

__global__ __launch_bounds__(FC1_THREADS) void dec_fc1_kernel(const DecFc1Launch L) {
  __shared__ __attribute__((aligned(16))) float Ht[FC1_ROWS * FC1_LD];              // hidden rows
  __shared__ __attribute__((aligned(16))) float Dx[FC1_ROWS * FC1_LD];              // d x_hat of this column group
  __shared__ __attribute__((aligned(16))) float Pt[(FC1_WAVES / 2) * FC1_ROWS * FC1_LD];  // partial dH tiles (two waves each)
  __shared__ float red[FC1_WAVES];
  LSTAMP(2, 0);
  // which decoder, row tile, column group
  int m = 0;
#pragma unroll
  for (int i = 1; i < 3; ++i)
    if (i < L.n_items && (int)blockIdx.x >= L.it[i].tile_begin) m = i;
  const DecFc1Item& I = L.it[m];
  const int d = I.d, h = I.h, Hp = I.Hp;
  const int local = (int)blockIdx.x - I.tile_begin;
  const int cg = local % I.col_groups;
  const int row0 = (local / I.col_groups) * FC1_ROWS;
  const int NF1 = (d + 15) >> 4;
  const int f0 = cg * I.frags_per_group, nfw = min(I.frags_per_group, NF1 - f0);      // this group's fragments
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bi = lane & 15, q = lane >> 4;
  const int J = Hp >> 4;                          // 16-wide reduction blocks of product 1 == output fragments of product 2

  // ---- requests.  The 16 hidden rows (pad units of the saved states are exact zeros): one 16-byte piece per thread
  const int per_row = Hp >> 2;                    // 16 * per_row <= 512
  const int hr = tid / per_row, hk = (tid - hr * per_row) << 2;
  f32x4 hreg = f32x4{0.f, 0.f, 0.f, 0.f};
  if (hr < FC1_ROWS && row0 + hr < L.rows) hreg = *reinterpret_cast<const f32x4*>(I.hs + (int64_t)(row0 + hr) * Hp + hk);
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)I.w, 0, d * h * 4, 0x00020000);
  const int n = (f0 + wave) * 16 + bi;            // this lane's output column in product 1
  const bool cok = (int)(wave < nfw) & (int)(n < d);
  f32x4 w1[FC1_MAXF];
#pragma unroll
  for (int j = 0; j < FC1_MAXF; ++j) {
    const bool ok = (int)cok & (int)(j < J);
    w1[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, ok ? (n * h + 16 * j + 4 * q) * 4 : FC1_OOB, 0, 0));
  }
  const int KS = nfw * 4;                         // 4-wide reduction steps of product 2 over this group's columns
  const int per = (KS + FC1_WAVES - 1) / FC1_WAVES;     // <= FC1_KS2W
  const int ks0 = wave * per;
  float w2[FC1_KS2W][FC1_MAXF];
#pragma unroll
  for (int i = 0; i < FC1_KS2W; ++i) {
    const int k = f0 * 16 + 4 * (ks0 + i) + q;    // column of Wfc^T == row of Wfc
#pragma unroll
    for (int f = 0; f < FC1_MAXF; ++f) {
      const int c = f * 16 + bi;
      const bool ok = (int)(L.with_bwd != 0) & (int)(i < per) & (int)(ks0 + i < KS) & (int)(k < d) & (int)(c < h);
      w2[i][f] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wres, ok ? (k * h + c) * 4 : FC1_OOB, 0, 0));
    }
  }
  float xv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + 4 * q + r;
    xv[r] = (cok && row < L.rows) ? I.x[(int64_t)row * I.ldx + n] : 0.0f;
  }
  const float bv = cok ? I.bias[n] : 0.0f;
  const bool rb = L.bf16 != 0;
  if (rb) {
#pragma unroll
    for (int e = 0; e < 4; ++e) hreg[e] = rnd_bf16(hreg[e], true);
  }
  if (hr < FC1_ROWS) *reinterpret_cast<f32x4*>(Ht + hr * FC1_LD + hk) = hreg;
  lds_barrier();
  LSTAMP(2, 1);

  // ---- product 1 (branch-free: blocks >= J were requested as zeros and meet a clamped hidden block; units >= h of a
  //      weight row belong to the next row -- finite values -- and meet zeros of the hidden tile)
  f32x4 acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
  if (wave < nfw) {
#pragma unroll
    for (int j = 0; j < FC1_MAXF; ++j) {
      const f32x4 hv = *reinterpret_cast<const f32x4*>(Ht + bi * FC1_LD + 16 * min(j, J - 1) + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc1 = mma16x16x4(hv[e], rnd_bf16(w1[j][e], rb), acc1);
    }
  }
  // ---- squared-error epilogue
  float lsum = 0.0f;
  if (wave < nfw) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * q + r;
      float dx = 0.0f;
      if (cok && row0 + row < L.rows) {
        const float xh = acc1[r] + bv;
        const float diff = xh - xv[r];
        lsum = fmaf(diff, diff, lsum);
        dx = I.grad_scale * diff;
        const int64_t o = (int64_t)(row0 + row) * d + n;
        if (I.xhat) I.xhat[o] = xh;
        if (I.dxhat) I.dxhat[o] = dx;
      }
      Dx[row * FC1_LD + wave * 16 + bi] = rnd_bf16(dx, rb);
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) lsum += __shfl_xor(lsum, o, 64);
  if (lane == 0) red[wave] = lsum;
  lds_barrier();
  if (tid == 0 && I.loss) {
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < FC1_WAVES; ++w) s += red[w];
    atomicAdd(I.loss, s * I.inv_count);
  }
  LSTAMP(2, 2);
  if (!L.with_bwd) return;

  // ---- product 2: dH[16, Hp] (+)= dx_hat[16, group columns] Wfc[group columns, h]
  {
    f32x4 acc2[FC1_MAXF];
#pragma unroll
    for (int f = 0; f < FC1_MAXF; ++f) acc2[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    // branch-free: masked weights are zeros, the dx_hat column index is clamped
#pragma unroll
    for (int i = 0; i < FC1_KS2W; ++i) {
      const float a = Dx[bi * FC1_LD + 4 * min(ks0 + i, KS - 1) + q];
#pragma unroll
      for (int f = 0; f < FC1_MAXF; ++f) acc2[f] = mma16x16x4(a, rnd_bf16(w2[i][f], rb), acc2[f]);
    }
    // waves 4..7 park their tiles, waves 0..3 add theirs on top (same lane -> same element), then the 4 tiles are summed
    float* P = Pt + (wave & 3) * FC1_ROWS * FC1_LD;
    if (wave >= 4) {
#pragma unroll
      for (int f = 0; f < FC1_MAXF; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) P[(4 * q + r) * FC1_LD + f * 16 + bi] = acc2[f][r];
    }
    lds_barrier();
    if (wave < 4) {
#pragma unroll
      for (int f = 0; f < FC1_MAXF; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) P[(4 * q + r) * FC1_LD + f * 16 + bi] += acc2[f][r];
    }
  }
  lds_barrier();
  if (hr < FC1_ROWS && row0 + hr < L.rows) {
    f32x4 s = *reinterpret_cast<const f32x4*>(Pt + hr * FC1_LD + hk);
#pragma unroll
    for (int w = 1; w < FC1_WAVES / 2; ++w) s += *reinterpret_cast<const f32x4*>(Pt + (w * FC1_ROWS + hr) * FC1_LD + hk);
    float* o = I.dhs + (int64_t)(row0 + hr) * Hp + hk;      // pad units: exact zeros (masked weights)
    if (I.col_groups == 1) {
      *reinterpret_cast<f32x4*>(o) = s;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(o + e, s[e]);
    }
  }
  LSTAMP_W(2, 15);
}

// MFM_ERR_UNSUPPORTED: a shape this kernel does not take (the caller falls back to the two GEMM launches)
// `dhs_zeroed`: the caller has put the dH buffers into the step's zero spans (required for more than one column group)
int dec_fc1_launch(DecFc1Launch& L, bool dhs_zeroed, hipStream_t stream) {
  MFM_REQUIRE(L.n_items >= 1 && L.n_items <= 3 && L.rows >= 1, "dec fc1: bad launch");
  int tiles = 0;
  const int row_tiles = (L.rows + FC1_ROWS - 1) / FC1_ROWS;
  for (int i = 0; i < L.n_items; ++i) {
    DecFc1Item& I = L.it[i];
    MFM_REQUIRE(I.hs && I.w && I.bias && I.x && I.d >= 1 && I.h >= 1 && I.Hp >= I.h && (I.Hp & 15) == 0, "dec fc1: item %d", i);
    MFM_REQUIRE(!L.with_bwd || I.dhs, "dec fc1: item %d: backward without a dH buffer", i);
    if (I.Hp > 16 * FC1_MAXF || (int64_t)I.d * I.h >= ((int64_t)1 << 28)) return MFM_ERR_UNSUPPORTED;
    const int NF1 = (I.d + 15) >> 4;
    // (rounds 3-5: 5 fragments per group, "spread the columns" -- decoder l on 160 workgroups.  Round 6, the launch clock: the
    // partial dH tiles of a row tile's column groups are added with memory-side atomics, and those, not the products, end the
    // launch: the single-group tiles of the narrow decoders leave 2.4 us after their first product, the four-group tiles 3.9
    // (median) to 5.4 us.  Step time against fragments per group: 3 0.1484, 4 0.1467, 5 0.1454, 7 0.1453, 8 0.1442 ms; ten waves
    // with groups of 10 were built and are slower, 0.1467.  MFM_FC1_FPG overrides.)
    int fpg = FC1_MAXF;
    if (const char* e = opt_get("MFM_FC1_FPG")) { const int v = atoi(e); if (v >= 1 && v <= FC1_MAXF && dhs_zeroed) fpg = v; }
    if (NF1 > fpg && !(dhs_zeroed || !L.with_bwd)) return MFM_ERR_UNSUPPORTED;      // several groups add into dH
    I.frags_per_group = std::min(fpg, NF1);
    I.col_groups = (NF1 + I.frags_per_group - 1) / I.frags_per_group;
    I.tile_begin = tiles;
    tiles += row_tiles * I.col_groups;
  }
  LSTAMP_BIND();
  MFM_LAUNCH_TIMED(dec_fc1_kernel, dim3(tiles), dim3(FC1_THREADS), 0, stream, L);
  MFM_LAUNCH_CHECK("dec_fc1_kernel");
  return MFM_OK;
}

}  // namespace mfm
