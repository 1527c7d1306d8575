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
#include <string.h>

#include <algorithm>

#include "internal.h"
#include "pack_dev.h"

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
constexpr int FL_LDW = FC1_LDW;      // bf16 elements per row of the W image / the H tile (128 + 8: pad of 16 bytes; pack_dev.h)

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


// ---------------------------------------------------------------------------------------------------------------------
// 64-row tiles (the default).  The 16-row kernel above pays five barriers and two dependent LDS hand-overs per 16 rows
// (~5 us per tile against ~1 us of work: 93 us at T*B = 40960); here a tile is 64 rows, there are two barriers per tile,
// and both products are taken TRANSPOSED so that an accumulator lane holds 4 consecutive columns of one row:
//   product 1: D[n][row] = W[n][:] . H[row][:]  (A operand = rows of the W image, B operand = rows of the H tile): the lane
//              gets x_hat[row][n0 .. n0+3] -- one 16-byte load of the target, one 8-byte store of d x_hat (global and LDS);
//   product 2: D[k][row] = sum_n W[n][k] dx_hat[row][n]  (A operand = the transposing read of the W image that used to be
//              the B operand, B operand = the dx_hat rows): the lane gets dH[row][k0 .. k0+3] -- one 8-byte store.  The 8
//              waves split the OUTPUT (4 row fragments x 2 halves of the hidden fragments) instead of the reduction, so no
//              partial tiles cross LDS.
constexpr int FL_R64 = 64;

template <int HP16>
__device__ __forceinline__ void fc1_large64_body(const DecFc1LargeLaunch& L, const DecFc1LargeItem& I, unsigned char* smem, float* red) {
  // (the hidden size is a template parameter: with run-time trip counts every MFMA of the two products sits behind a branch)
  constexpr int KB1 = (HP16 + 1) / 2, J2 = HP16, JH = (HP16 + 1) / 2, Hp = 16 * HP16;
  const int d = I.d, h = I.h;
  const int wg = (int)blockIdx.x - I.wg_begin, nwg = I.wg_count;
  const int NF1 = (d + 15) >> 4;
  const int NB2 = (NF1 * 16 + 31) >> 5;
  const int LDX = NB2 * 32 + 8;
  __bf16* Wb = reinterpret_cast<__bf16*>(smem);                               // [NB2 * 32][FL_LDW]
  __bf16* Ht = Wb + (size_t)NB2 * 32 * FL_LDW;                                // [64][FL_LDW]
  __bf16* Dx = Ht + FL_R64 * FL_LDW;                                          // [64][LDX]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bi = lane & 15, q = lane >> 4;

  if (I.wimg) {
    // the image was packed once for the whole launch (fc1_pack_kernel): 16-byte copies, all requests of a thread in flight
    const f32x4* src = reinterpret_cast<const f32x4*>(I.wimg);
    f32x4* dst = reinterpret_cast<f32x4*>(Wb);
    const int n16 = NB2 * 32 * FL_LDW / 8;
    for (int base = tid; base < n16; base += FL_THREADS * 6) {
      f32x4 v[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) v[u] = src[min(base + u * FL_THREADS, n16 - 1)];
#pragma unroll
      for (int u = 0; u < 6; ++u)
        if (base + u * FL_THREADS < n16) dst[base + u * FL_THREADS] = v[u];
    }
  } else {
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
  }
  {   // zero the two row tiles once (pad columns never receive data): Ht and Dx are contiguous, 16-byte shaped
    f32x4* z = reinterpret_cast<f32x4*>(Ht);
    const int n16 = (FL_R64 * FL_LDW + FL_R64 * LDX) / 8;
    for (int idx = tid; idx < n16; idx += FL_THREADS) z[idx] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // per-lane constants of product 1's epilogue: output fragment f = wave + 8 i, columns n0 .. n0 + 3
  f32x4 bv[FL_MAXF];
  int n0[FL_MAXF];
#pragma unroll
  for (int i = 0; i < FL_MAXF; ++i) {
    n0[i] = (wave + FL_WAVES * i) * 16 + 4 * q;
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[i][r] = (n0[i] + r < d) ? I.bias[n0[i] + r] : 0.0f;
  }
  __syncthreads();

  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  constexpr unsigned FL_OOB = 0x7FFFFFF0u;
  const __amdgpu_buffer_rsrc_t hres = __builtin_amdgcn_make_buffer_rsrc((void*)I.hs, 0, (int)min((int64_t)L.rows * Hp * 2, (int64_t)0x7FFFFF00), 0x00020000);
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)I.x, 0, (int)min(((int64_t)(L.rows - 1) * I.ldx + d) * 4, (int64_t)0x7FFFFF00), 0x00020000);
  const __amdgpu_buffer_rsrc_t dxres = __builtin_amdgcn_make_buffer_rsrc(I.dxhat, 0, (int)min((int64_t)L.rows * I.ld_dxhat * 2, (int64_t)0x7FFFFF00), 0x00020000);
  const __amdgpu_buffer_rsrc_t dhres = __builtin_amdgcn_make_buffer_rsrc(I.dhs, 0, (int)min((int64_t)L.rows * Hp * 2, (int64_t)0x7FFFFF00), 0x00020000);
  const int n_tiles = (L.rows + FL_R64 - 1) / FL_R64;
  const int per_row = Hp >> 3;                      // 16-byte pieces per hidden row: 64 rows = at most 1024 pieces, 2 per thread
  int hr[2], hk[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int c = tid + j * FL_THREADS; hr[j] = c / per_row; hk[j] = (c - hr[j] * per_row) << 3; }
  const int rrow = 4 * q + (bi >> 2), rcol = 4 * (bi & 3);      // transposing-read coordinates
  float lsum = 0.0f;
  // d x_hat leaves from its LDS tile as whole 16-byte pieces in memory order (round 5): an accumulator lane holds 4 columns of
  // one row, so the direct store touched 16 rows per instruction -- 12 instructions per wave and tile, 1,536 cache-line visits
  // for a tile of ~300 lines, ~3 cycles each on the CU's address path (profiles/r05_seq_bf16_study.txt section 7)
  constexpr int FL_DXP = 6;                         // pieces per thread: 64 rows x ld / 8 <= 6 x 512 (ld <= 384)
  const int dx_ppr = I.ld_dxhat >> 3;               // 16-byte pieces per row (ld is a multiple of 8 here: see the launcher)
  int dxp_lds[FL_DXP], dxp_g[FL_DXP], dxp_row[FL_DXP];
#pragma unroll
  for (int j = 0; j < FL_DXP; ++j) {
    const int c = tid + j * FL_THREADS;
    const int r = c / dx_ppr, k8 = (c - r * dx_ppr) << 3;
    dxp_row[j] = r < FL_R64 ? r : 0x40000000;       // (beyond the tile: never below L.rows)
    dxp_lds[j] = (r < FL_R64 ? r * LDX + min(k8, LDX - 8) : 0) * 2;
    dxp_g[j] = (r * I.ld_dxhat + k8) * 2;
  }

  // every global access is a buffer instruction whose offset is out of range for lanes without data: nothing sits under a
  // lane-divergent branch, and the next tile's hidden rows and targets are requested while the current tile is multiplied
  f32x4 hraw[2];
  f32x4 xv[FL_MAXF][4];
  auto request_h = [&](int tile) {
    const int row0 = tile * FL_R64;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      hraw[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
          hres, (hr[j] < FL_R64 && row0 + hr[j] < L.rows) ? (unsigned)(((row0 + hr[j]) * Hp + hk[j]) * 2) : FL_OOB, 0, 0));
  };
  auto request_x = [&](int tile, int i, int rt) {
    const int row = tile * FL_R64 + rt * 16 + bi;
    // (a 16-byte load whose tail leaves the row reads the next row's first columns, or 0 behind the buffer: masked below)
    const bool ok = (wave + FL_WAVES * i < NF1) && n0[i] < d && row < L.rows;
    xv[i][rt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, ok ? (unsigned)((row * (int)I.ldx + n0[i]) * 4) : FL_OOB, 0, 0));
  };
  request_h(wg);
#pragma unroll
  for (int i = 0; i < FL_MAXF; ++i)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) request_x(wg, i, rt);

  for (int tile = wg; tile < ((L.dbg & 4) ? 0 : n_tiles); tile += nwg) {
    const int row0 = tile * FL_R64;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (hr[j] < FL_R64) *reinterpret_cast<f32x4*>(Ht + hr[j] * FL_LDW + hk[j]) = hraw[j];
    request_h(tile + nwg);
    lds_barrier();                                  // the H tile is in place; everybody is done with the previous tile's dx_hat
    // ---- product 1
    if (!(L.dbg & 1)) {
      bf16x8 wf[FL_MAXF][KB1];
#pragma unroll
      for (int i = 0; i < FL_MAXF; ++i)
#pragma unroll
        for (int kb = 0; kb < KB1; ++kb)
          wf[i][kb] = (wave + FL_WAVES * i < NF1)
                          ? *reinterpret_cast<const bf16x8*>(Wb + (size_t)((wave + FL_WAVES * i) * 16 + bi) * FL_LDW + kb * 32 + 8 * q)
                          : bf16x8{};
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        bf16x8 hf[KB1];
#pragma unroll
        for (int kb = 0; kb < KB1; ++kb) hf[kb] = *reinterpret_cast<const bf16x8*>(Ht + (rt * 16 + bi) * FL_LDW + kb * 32 + 8 * q);
        const int row = row0 + rt * 16 + bi;
#pragma unroll
        for (int i = 0; i < FL_MAXF; ++i) {
          if (wave + FL_WAVES * i < NF1) {          // wave-uniform
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KB1; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][kb], hf[kb], acc, 0, 0, 0);
            // accumulator lane (bi, q): x_hat[row][n0 + r]
            const f32x4 xt = xv[i][rt];
            f32x4 dx;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const bool ok = n0[i] + r < d && row < L.rows;
              const float diff = ok ? acc[r] + bv[i][r] - xt[r] : 0.0f;
              lsum = fmaf(diff, diff, lsum);
              dx[r] = I.grad_scale * diff;
            }
            request_x(tile + nwg, i, rt);           // (the register is free again: the next tile's target)
            const bf16x4 dxb = __builtin_convertvector(dx, bf16x4);
            *reinterpret_cast<bf16x4*>(Dx + (rt * 16 + bi) * LDX + n0[i]) = dxb;
          }
        }
      }
    }
    lds_barrier();
    // the tile's d x_hat to memory (the next tile's product 1 overwrites Dx only behind the next barrier)
    if (!(L.dbg & 1)) {
      u32x4 pv[FL_DXP];
#pragma unroll
      for (int j = 0; j < FL_DXP; ++j) pv[j] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(Dx) + dxp_lds[j]);
#pragma unroll
      for (int j = 0; j < FL_DXP; ++j)
        __builtin_amdgcn_raw_buffer_store_b128(pv[j], dxres, (row0 + dxp_row[j] < L.rows) ? (unsigned)(row0 * I.ld_dxhat * 2 + dxp_g[j]) : FL_OOB, 0, 0);
    }
    // ---- product 2: wave (rt, half) owns dH fragments rows 16 rt .., hidden fragments half * JH .. of the tile
    if (!(L.dbg & 2)) {
      const int rt = wave & 3, half = wave >> 2;
      f32x4 acc2[JH];
#pragma unroll
      for (int j = 0; j < JH; ++j) acc2[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int nb = 0; nb < NB2; ++nb) {
        const __bf16* ap = Dx + (rt * 16 + bi) * LDX + nb * 32 + 4 * q;
        const bf16x8 a = cat8(*reinterpret_cast<const bf16x4*>(ap), *reinterpret_cast<const bf16x4*>(ap + 16));
#pragma unroll
        for (int j = 0; j < JH; ++j)
          if (half * JH + j < J2) {          // wave-uniform (odd fragment counts: the second half has one less)
            const __bf16* bp = Wb + (size_t)(nb * 32 + rrow) * FL_LDW + (half * JH + j) * 16 + rcol;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)bp);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(bp + 16 * FL_LDW));
            const bf16x8 b = cat8(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi));
            acc2[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc2[j], 0, 0, 0);
          }
      }
      // accumulator lane (bi, q): dH[row0 + 16 rt + bi][16 f2 + 4q + r]; pad units (>= h) come out as exact zeros
      const int row = row0 + rt * 16 + bi;
#pragma unroll
      for (int j = 0; j < JH; ++j)
        if (half * JH + j < J2) {
          const bf16x4 o = __builtin_convertvector(acc2[j], bf16x4);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), dhres,
                                                row < L.rows ? (unsigned)((row * Hp + (half * JH + j) * 16 + 4 * q) * 2) : FL_OOB, 0, 0);
        }
    }
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



__global__ __launch_bounds__(FL_THREADS) void dec_fc1_large64_kernel(const DecFc1LargeLaunch L) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[FL_WAVES];
  int m = 0;
#pragma unroll
  for (int i = 1; i < 3; ++i)
    if (i < L.n_items && (int)blockIdx.x >= L.it[i].wg_begin) m = i;
  const DecFc1LargeItem& I = L.it[m];
  switch (I.Hp >> 4) {
    case 1: fc1_large64_body<1>(L, I, smem, red); break;
    case 2: fc1_large64_body<2>(L, I, smem, red); break;
    case 3: fc1_large64_body<3>(L, I, smem, red); break;
    case 4: fc1_large64_body<4>(L, I, smem, red); break;
    case 5: fc1_large64_body<5>(L, I, smem, red); break;
    case 6: fc1_large64_body<6>(L, I, smem, red); break;
    case 7: fc1_large64_body<7>(L, I, smem, red); break;
    default: fc1_large64_body<8>(L, I, smem, red); break;
  }
}

// the bf16 image of Wfc the workgroups keep in LDS: [NB2 * 32][FL_LDW], rows n >= d and columns k >= h zero (pack_dev.h)
__global__ __launch_bounds__(256) void fc1_pack_kernel(const Fc1PackArgs A) {
  fc1_pack_body(A, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

}  // namespace

int dec_fc1_large_supported(const DecFc1LargeItem& I) {
  if (!I.hs || !I.w || !I.bias || !I.x || !I.dxhat || !I.dhs) return 0;
  if (I.d < 1 || I.h < 1 || I.Hp < I.h || (I.Hp & 15) || I.Hp > 32 * FL_MAXKB) return 0;
  if ((I.d + 15) / 16 > FL_WAVES * FL_MAXF) return 0;
  if (I.ld_dxhat < I.d) return 0;
  return 1;
}

size_t dec_fc1_large_wimg_bytes(int d) {
  const int NF1 = (d + 15) / 16, NB2 = (NF1 * 16 + 31) / 32;
  return (size_t)NB2 * 32 * FL_LDW * 2;
}

// columns of the d x_hat tile in LDS (the 64-row kernel copies ld_dxhat of them per row to memory)
static int fl_dx_cols(const DecFc1LargeItem& I) { return ((I.d + 15) / 16 * 16 + 31) / 32 * 32 + 8; }
static size_t fl_lds_bytes(const DecFc1LargeItem& I, int rows_per_tile) {
  const int NF1 = (I.d + 15) / 16, NB2 = (NF1 * 16 + 31) / 32;
  if (rows_per_tile == FL_R64) return ((size_t)NB2 * 32 * FL_LDW + (size_t)FL_R64 * FL_LDW + (size_t)FL_R64 * (NB2 * 32 + 8)) * 2;
  return ((size_t)NB2 * 32 * FL_LDW + (size_t)FL_ROWS * FL_LDW + (size_t)FL_ROWS * (NB2 * 32 + 8)) * 2 + (size_t)4 * FL_ROWS * FL_LDW * 4;
}

// the pack arguments of a launch's items (those that carry a wimg scratch)
int fc1_pack_prepare(const DecFc1LargeLaunch& L, Fc1PackArgs* out) {
  Fc1PackArgs& A = *out;
  memset(&A, 0, sizeof(A));
  int at = 0;
  for (int i = 0; i < L.n_items; ++i) {
    const DecFc1LargeItem& I = L.it[i];
    if (!I.wimg) continue;
    MFM_REQUIRE((((uintptr_t)I.wimg) & 15) == 0, "dec fc1 (large): item %d: weight image scratch not 16-byte aligned", i);
    const int rows = (int)(dec_fc1_large_wimg_bytes(I.d) / (FL_LDW * 2));
    A.w[A.n] = I.w; A.out[A.n] = reinterpret_cast<__bf16*>(I.wimg); A.d[A.n] = I.d; A.h[A.n] = I.h; A.rows[A.n] = rows;
    A.begin[A.n] = at; at += rows * (FL_LDW / 8);
    ++A.n;
  }
  for (int i = A.n; i < 4; ++i) A.begin[i] = at;
  return MFM_OK;
}

// true when dec_fc1_large_launch would take the 64-row kernel (the only one that reads the packed images)
bool dec_fc1_large_uses_wimg(const DecFc1LargeLaunch& L) {
  if (opt_get("MFM_FC1_LARGE_ROWS") && atoi(opt_get("MFM_FC1_LARGE_ROWS")) == 16) return false;
  for (int i = 0; i < L.n_items; ++i)
    if (fl_lds_bytes(L.it[i], FL_R64) > 156 * 1024 || (L.it[i].ld_dxhat & 7) || L.it[i].ld_dxhat > fl_dx_cols(L.it[i]) || (L.it[i].Hp >> 3) > 16 ||
        (int64_t)L.rows * L.it[i].ldx >= ((int64_t)1 << 29))
      return false;
  return true;
}

int dec_fc1_large_launch(DecFc1LargeLaunch& L, hipStream_t stream) {
  MFM_REQUIRE(L.n_items >= 1 && L.n_items <= 3 && L.rows >= 1, "dec fc1 (large): bad launch");
  // 64-row tiles unless they do not fit the LDS (or MFM_FC1_LARGE_ROWS=16 asks for the 16-row kernel)
  int RT = (opt_get("MFM_FC1_LARGE_ROWS") && atoi(opt_get("MFM_FC1_LARGE_ROWS")) == 16) ? FL_ROWS : FL_R64;
  for (int i = 0; i < L.n_items; ++i) {
    MFM_REQUIRE(dec_fc1_large_supported(L.it[i]), "dec fc1 (large): item %d is not supported", i);
    if (RT == FL_R64 && (fl_lds_bytes(L.it[i], FL_R64) > 156 * 1024 || (L.it[i].ld_dxhat & 7) || L.it[i].ld_dxhat > fl_dx_cols(L.it[i]) || (L.it[i].Hp >> 3) > 16 ||
                         (int64_t)L.rows * L.it[i].ldx >= ((int64_t)1 << 29)))
      RT = FL_ROWS;
  }
  L.dbg = opt_get("MFM_FC1_LARGE_DBG") ? atoi(opt_get("MFM_FC1_LARGE_DBG")) : 0;
  const int n_tiles = (L.rows + RT - 1) / RT;
  // one workgroup per CU; workgroups per decoder in proportion to its cost per row tile: a fixed part (barriers, the
  // tile's loads) plus the matrix part
  const double c0_def = RT == FL_R64 ? 0.25 : 0.5;      // 16-row tiles, measured at B = 2048: 0.25 -> 124, 0.5 -> 93, 1 -> 94, 2 -> 106, 8 -> 119 us
  const double c0 = opt_get("MFM_FC1_LARGE_C0") ? atof(opt_get("MFM_FC1_LARGE_C0")) : c0_def;      // tuning override
  auto cost = [c0](const DecFc1LargeItem& I) { return c0 + (double)I.d * I.h / 31200.0; };
  double wsum = 0.0;
  size_t smem = 0;
  for (int i = 0; i < L.n_items; ++i) {
    wsum += cost(L.it[i]);
    smem = std::max(smem, fl_lds_bytes(L.it[i], RT));
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
    MFM_HIP_CHECK(hipFuncSetAttribute((const void*)dec_fc1_large64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    attr = true;
  }
  if (RT == FL_R64) {
    if (!L.packed) {          // (the plan packs the images with the step's other weight images: pack_all_launch)
      Fc1PackArgs A;
      const int rc = fc1_pack_prepare(L, &A);
      if (rc != MFM_OK) return rc;
      if (A.n > 0) {
        MFM_LAUNCH_TIMED(fc1_pack_kernel, dim3((A.begin[A.n] + 255) / 256), dim3(256), 0, stream, A);
        MFM_LAUNCH_CHECK("fc1_pack_kernel");
      }
    }
  } else {
    for (int i = 0; i < L.n_items; ++i) L.it[i].wimg = nullptr;
  }
  if (RT == FL_R64) MFM_LAUNCH_TIMED(dec_fc1_large64_kernel, dim3(total), dim3(FL_THREADS), smem, stream, L);
  else MFM_LAUNCH_TIMED(dec_fc1_large_kernel, dim3(total), dim3(FL_THREADS), smem, stream, L);
  MFM_LAUNCH_CHECK("dec_fc1_large_kernel");
  return MFM_OK;
}

}  // namespace mfm
