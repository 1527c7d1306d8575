// Decoder output layer of a bf16-RESIDENT plan at LARGE T*B: forward, squared error, d x_hat and backward-to-hidden in ONE
// launch of persistent workgroups (round 3).
//
// Reference: decoderLSTM.fc1 over every decoder hidden state (mfm_model.py:88-90), `lda_x* * mse(x_hat, x)`
// (mfm_mosi.py:437) and what autograd sends back through fc1 to the hidden states:
//     x_hat = H Wfc^T + b      diff = x_hat - x      loss += sum diff^2 / count      dx_hat = 2 lda / count * diff
//     dH    = dx_hat Wfc
// On the bf16-resident plan these were two grouped bf16 GEMM launches (fc1 + squared-error epilogue 113 us, dH 54 us at
// B = 2048: profiles/r03_bf16_resident_experiments.txt) whose K loops are 2-5 tiles long -- all prologue and epilogue: 4-byte
// target loads and 2-byte d x_hat stores per accumulator element, and every 64 x 64 tile re-reads its weights.  The launch
// moves ~110 MB (H bf16 in, the fp32 targets in, dx_hat and dH bf16 out) and needs 5 GFLOP: it should cost what that
// traffic costs.
//
// Here a workgroup keeps ONE bf16 image of Wfc [d, h] in LDS for its whole life and walks 16-row tiles of its decoder:
//   product 1 (x_hat^T tile by tile): wave w owns output fragments w, w + 8, w + 16; A fragment = 8 consecutive hidden
//             units of a row of the H tile, B fragment = 8 consecutive hidden units of a row of W: both plain 16-byte LDS reads;
//   product 2 (dH = dx_hat W): the reduction runs over the OUTPUT columns n, i.e. down the rows of the same W image -- the B
//             fragment is read with ds_read_b64_tr_b16 (lane (bi, q) receives rows 4q .. 4q+3 of column bi; two reads make
//             the 8-deep k slice), the A fragment (dx_hat tile, staged in LDS as bf16) with two 8-byte reads of the same k
//             assignment; the 32-row blocks of n are dealt round-robin to the 8 waves, partial dH tiles are summed through LDS.
// Operands are the bf16 values the GEMM path uses (H as stored, W and dx_hat rounded to nearest even), accumulation is fp32
// on v_mfma_f32_16x16x32_bf16: the same arithmetic, the parity gates of the bf16 suites apply unchanged.
// The weight-gradient products dWfc = dx_hat^T H and dbfc ride in the one-pass launch (dw_bf16.hip).
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <algorithm>

#include "internal.h"

namespace mfm {

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int FL_THREADS = 512;
constexpr int FL_WAVES = 8;
constexpr int FL_ROWS = 16;
constexpr int FL_MAXKB = 4;          // hidden size <= 128: k-blocks of 32 in product 1, output fragments <= 8 in product 2
constexpr int FL_MAXF = 3;           // output fragments per wave in product 1: d <= 16 * 8 * 3 = 384
constexpr int FL_LDW = 128 + 8;      // bf16 elements per row of the W image / the H tile (pad: 16 bytes)

__device__ __forceinline__ bf16x8 cat8(bf16x4 a, bf16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }

__global__ __launch_bounds__(FL_THREADS) void dec_fc1_large_kernel(const DecFc1LargeLaunch L) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // which decoder / which workgroup of it
  int m = 0;
#pragma unroll
  for (int i = 1; i < 3; ++i)
    if (i < L.n_items && (int)blockIdx.x >= L.it[i].wg_begin) m = i;
  const DecFc1LargeItem& I = L.it[m];
  const int d = I.d, h = I.h, Hp = I.Hp;
  const int wg = (int)blockIdx.x - I.wg_begin, nwg = I.wg_count;
  const int NF1 = (d + 15) >> 4;                  // output fragments
  const int DP = NF1 * 16;                         // padded output columns
  const int NB2 = (DP + 31) >> 5;                  // 32-column reduction blocks of product 2
  const int KB1 = (Hp + 31) >> 5;                  // 32-unit reduction blocks of product 1
  const int J2 = Hp >> 4;                          // output fragments of product 2
  const int LDX = NB2 * 32 + 8;                    // bf16 elements per row of the dx_hat tile
  __bf16* Wb = reinterpret_cast<__bf16*>(smem);                               // [NB2 * 32][FL_LDW]
  __bf16* Ht = Wb + (size_t)NB2 * 32 * FL_LDW;                                // [16][FL_LDW]
  __bf16* Dx = Ht + FL_ROWS * FL_LDW;                                         // [16][LDX]
  float* Pt = reinterpret_cast<float*>(Dx + FL_ROWS * LDX);                   // [4][16][FL_LDW] partial dH tiles (16-byte aligned: all counts are multiples of 8)
  __shared__ float red[FL_WAVES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bi = lane & 15, q = lane >> 4;

  // ---- the W image: rows n < d, columns k < h from the fp32 master copy, everything else zero
  for (int idx = tid; idx < NB2 * 32 * (FL_LDW / 4); idx += FL_THREADS) {
    const int n = idx / (FL_LDW / 4), k4 = (idx - n * (FL_LDW / 4)) * 4;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (n < d) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k4 + e < h) v[e] = I.w[(int64_t)n * h + k4 + e];
    }
    *reinterpret_cast<bf16x4*>(Wb + (size_t)n * FL_LDW + k4) = __builtin_convertvector(v, bf16x4);
  }
  // zero the pad columns of the two row tiles once (k >= Hp of H, n >= DP of dx_hat never receive data)
  for (int idx = tid; idx < FL_ROWS * FL_LDW; idx += FL_THREADS) Ht[idx] = (__bf16)0.0f;
  for (int idx = tid; idx < FL_ROWS * LDX; idx += FL_THREADS) Dx[idx] = (__bf16)0.0f;
  // per-lane constants of product 1's epilogue
  float bv[FL_MAXF];
  int ncol[FL_MAXF];
#pragma unroll
  for (int i = 0; i < FL_MAXF; ++i) {
    const int f = wave + FL_WAVES * i;
    ncol[i] = f * 16 + bi;
    bv[i] = (f < NF1 && ncol[i] < d) ? I.bias[ncol[i]] : 0.0f;
  }
  __syncthreads();

  const __bf16* hs = reinterpret_cast<const __bf16*>(I.hs);
  __bf16* dxo = reinterpret_cast<__bf16*>(I.dxhat);
  __bf16* dho = reinterpret_cast<__bf16*>(I.dhs);
  const int n_tiles = (L.rows + FL_ROWS - 1) / FL_ROWS;
  const int per_row = Hp >> 3;                     // 16-byte pieces per hidden row (16 * per_row <= 256)
  const int hr = tid / per_row, hk = (tid - hr * per_row) << 3;
  const int rrow = 4 * q + (bi >> 2), rcol = 4 * (bi & 3);      // transposing-read coordinates
  float lsum = 0.0f;

  // Global traffic of the tile loop: the NEXT tile's hidden rows and targets are requested while the current tile is
  // multiplied (a tile otherwise starts with two dependent round trips to HBM, ~2 us each, against ~1 us of work), and every
  // load / store is a buffer instruction whose offset is out of range for lanes without data -- nothing sits under a
  // branch, so the wait for the prefetched registers is a counted vmcnt, not vmcnt(0) behind the tile's own stores.
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  constexpr unsigned FL_OOB = 0x7FFFFFF0u;
  const __amdgpu_buffer_rsrc_t hres = __builtin_amdgcn_make_buffer_rsrc((void*)hs, 0, (int)min((int64_t)L.rows * Hp * 2, (int64_t)0x7FFFFF00), 0x00020000);
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)I.x, 0, (int)min(((int64_t)(L.rows - 1) * I.ldx + d) * 4, (int64_t)0x7FFFFF00), 0x00020000);
  const __amdgpu_buffer_rsrc_t dxres = __builtin_amdgcn_make_buffer_rsrc((void*)dxo, 0, (int)min((int64_t)L.rows * I.ld_dxhat * 2, (int64_t)0x7FFFFF00), 0x00020000);
  const __amdgpu_buffer_rsrc_t dhres = __builtin_amdgcn_make_buffer_rsrc((void*)dho, 0, (int)min((int64_t)L.rows * Hp * 2, (int64_t)0x7FFFFF00), 0x00020000);
  f32x4 raw_n;
  float xv_n[FL_MAXF][4];
  auto prefetch = [&](int tile) {
    const int row0 = tile * FL_ROWS;               // tile >= n_tiles: every row is out of range, everything reads 0
    raw_n = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
        hres, (hr < FL_ROWS && row0 + hr < L.rows) ? (unsigned)(((row0 + hr) * Hp + hk) * 2) : FL_OOB, 0, 0));
#pragma unroll
    for (int i = 0; i < FL_MAXF; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + 4 * q + r;
        const bool ok = (wave + FL_WAVES * i < NF1) && ncol[i] < d && row < L.rows;
        xv_n[i][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xres, ok ? (unsigned)((row * (int)I.ldx + ncol[i]) * 4) : FL_OOB, 0, 0));
      }
  };
  prefetch(wg);
  for (int tile = wg; tile < n_tiles; tile += nwg) {
    const int row0 = tile * FL_ROWS;
    // ---- the H tile (bf16 as stored) -> LDS; this tile's targets; the next tile's requests
    if (hr < FL_ROWS) *reinterpret_cast<f32x4*>(Ht + hr * FL_LDW + hk) = raw_n;
    float xv[FL_MAXF][4];
#pragma unroll
    for (int i = 0; i < FL_MAXF; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) xv[i][r] = xv_n[i][r];
    prefetch(tile + nwg);
    lds_barrier();
    // ---- product 1: x_hat fragments (rows of the tile x 16 output columns), reduction over the hidden units
    {
      bf16x8 af[FL_MAXKB];
#pragma unroll
      for (int kb = 0; kb < FL_MAXKB; ++kb)
        af[kb] = (kb < KB1) ? *reinterpret_cast<const bf16x8*>(Ht + bi * FL_LDW + kb * 32 + 8 * q) : bf16x8{};
#pragma unroll
      for (int i = 0; i < FL_MAXF; ++i) {
        const int f = wave + FL_WAVES * i;
        if (f < NF1) {                            // wave-uniform
          f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kb = 0; kb < FL_MAXKB; ++kb)
            if (kb < KB1) {
              const bf16x8 wf = *reinterpret_cast<const bf16x8*>(Wb + (size_t)(f * 16 + bi) * FL_LDW + kb * 32 + 8 * q);
              acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kb], wf, acc, 0, 0, 0);
            }
          // accumulator lane: rows 4q + r of the tile, column ncol[i]
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = row0 + 4 * q + r;
            float dx = 0.0f;
            if (ncol[i] < d && row < L.rows) {
              const float diff = acc[r] + bv[i] - xv[i][r];
              lsum = fmaf(diff, diff, lsum);
              dx = I.grad_scale * diff;
            }
            const __bf16 dxb = (__bf16)dx;
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, dxb), dxres,
                                                  (ncol[i] < d && row < L.rows) ? (unsigned)((row * I.ld_dxhat + ncol[i]) * 2) : FL_OOB, 0, 0);
            Dx[(4 * q + r) * LDX + ncol[i]] = dxb;
          }
        }
      }
    }
    lds_barrier();
    // ---- product 2: dH tile = dx_hat tile [16, DP] x W [DP, h]; 32-column blocks of the reduction dealt to the waves
    {
      f32x4 acc2[2 * FL_MAXKB];
#pragma unroll
      for (int f = 0; f < 2 * FL_MAXKB; ++f) acc2[f] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int nb = wave; nb < NB2; nb += FL_WAVES) {
        // A: row bi of the dx_hat tile, k = 32 nb + {4q .. 4q+3, 16 + 4q .. 16 + 4q+3} (the k assignment of the transposing read)
        const __bf16* ap = Dx + bi * LDX + nb * 32 + 4 * q;
        const bf16x8 a = cat8(*reinterpret_cast<const bf16x4*>(ap), *reinterpret_cast<const bf16x4*>(ap + 16));
#pragma unroll
        for (int f = 0; f < 2 * FL_MAXKB; ++f)
          if (f < J2) {
            const __bf16* bp = Wb + (size_t)(nb * 32 + rrow) * FL_LDW + f * 16 + rcol;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)bp);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(bp + 16 * FL_LDW));
            const bf16x8 b = cat8(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi));
            acc2[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc2[f], 0, 0, 0);
          }
      }
      // waves 4..7 park their tiles, waves 0..3 add theirs on top (same lane -> same element); then 4 tiles are summed
      float* P = Pt + (wave & 3) * FL_ROWS * FL_LDW;
      if (wave >= 4) {
#pragma unroll
        for (int f = 0; f < 2 * FL_MAXKB; ++f)
          if (f < J2)
#pragma unroll
            for (int r = 0; r < 4; ++r) P[(4 * q + r) * FL_LDW + f * 16 + bi] = acc2[f][r];
      }
      lds_barrier();
      if (wave < 4) {
#pragma unroll
        for (int f = 0; f < 2 * FL_MAXKB; ++f)
          if (f < J2)
#pragma unroll
            for (int r = 0; r < 4; ++r) P[(4 * q + r) * FL_LDW + f * 16 + bi] += acc2[f][r];
      }
    }
    lds_barrier();
    {
      const int hrc = min(hr, FL_ROWS - 1);
      f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        s0 += *reinterpret_cast<const f32x4*>(Pt + (w * FL_ROWS + hrc) * FL_LDW + hk);
        s1 += *reinterpret_cast<const f32x4*>(Pt + (w * FL_ROWS + hrc) * FL_LDW + hk + 4);
      }
      // pad units (columns >= h): the W image holds zeros there, so they come out as exact zeros
      const bf16x8 o8 = cat8(__builtin_convertvector(s0, bf16x4), __builtin_convertvector(s1, bf16x4));
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o8), dhres,
                                             (hr < FL_ROWS && row0 + hr < L.rows) ? (unsigned)(((row0 + hr) * Hp + hk) * 2) : FL_OOB, 0, 0);
    }
    lds_barrier();                                 // the tiles are rewritten by the next iteration
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) lsum += __shfl_xor(lsum, o, 64);
  if (lane == 0) red[wave] = lsum;
  __syncthreads();
  if (tid == 0 && I.loss) {
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < FL_WAVES; ++w) s += red[w];
    atomicAdd(I.loss, s * I.inv_count);
  }
}

}  // namespace

int dec_fc1_large_supported(const DecFc1LargeItem& I) {
  if (!I.hs || !I.w || !I.bias || !I.x || !I.dxhat || !I.dhs) return 0;
  if (I.d < 1 || I.h < 1 || I.Hp < I.h || (I.Hp & 15) || I.Hp > 32 * FL_MAXKB) return 0;
  if ((I.d + 15) / 16 > FL_WAVES * FL_MAXF) return 0;
  if (I.ld_dxhat < I.d) return 0;
  return 1;
}

static size_t fl_lds_bytes(const DecFc1LargeItem& I) {
  const int NF1 = (I.d + 15) / 16, NB2 = (NF1 * 16 + 31) / 32;
  return ((size_t)NB2 * 32 * FL_LDW + (size_t)FL_ROWS * FL_LDW + (size_t)FL_ROWS * (NB2 * 32 + 8)) * 2 + (size_t)4 * FL_ROWS * FL_LDW * 4;
}

int dec_fc1_large_launch(DecFc1LargeLaunch& L, hipStream_t stream) {
  MFM_REQUIRE(L.n_items >= 1 && L.n_items <= 3 && L.rows >= 1, "dec fc1 (large): bad launch");
  const int n_tiles = (L.rows + FL_ROWS - 1) / FL_ROWS;
  // one workgroup per CU; workgroups per decoder in proportion to its cost per row tile: a fixed part (five barriers, the
  // tile's loads) plus the matrix part, about equal to it for the 300 x 104 language decoder
  const double c0 = getenv("MFM_FC1_LARGE_C0") ? atof(getenv("MFM_FC1_LARGE_C0")) : 0.5;      // tuning override; measured at B = 2048: 0.25 -> 124, 0.5 -> 93, 1 -> 94, 2 -> 106, 8 -> 119 us
  auto cost = [c0](const DecFc1LargeItem& I) { return c0 + (double)I.d * I.h / 31200.0; };
  double wsum = 0.0;
  size_t smem = 0;
  for (int i = 0; i < L.n_items; ++i) {
    MFM_REQUIRE(dec_fc1_large_supported(L.it[i]), "dec fc1 (large): item %d is not supported", i);
    wsum += cost(L.it[i]);
    smem = std::max(smem, fl_lds_bytes(L.it[i]));
  }
  MFM_REQUIRE(smem <= 156 * 1024, "dec fc1 (large): %zu bytes of LDS", smem);
  const int cus = device_cus();
  int total = 0;
  for (int i = 0; i < L.n_items; ++i) {
    DecFc1LargeItem& I = L.it[i];
    int n = (int)(cus * cost(I) / wsum + 0.5);
    n = std::max(1, std::min(n, n_tiles));
    I.wg_begin = total; I.wg_count = n;
    total += n;
  }
  static bool attr = false;
  if (!attr) {
    MFM_HIP_CHECK(hipFuncSetAttribute((const void*)dec_fc1_large_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    attr = true;
  }
  hipLaunchKernelGGL(dec_fc1_large_kernel, dim3(total), dim3(FL_THREADS), smem, stream, L);
  MFM_LAUNCH_CHECK("dec_fc1_large_kernel");
  return MFM_OK;
}

}  // namespace mfm
