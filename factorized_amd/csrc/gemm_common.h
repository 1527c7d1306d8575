// Shared between the fp32 (gemm.hip) and bf16-operand (gemm_bf16.hip) grouped GEMM kernels: the problem table
// that travels in the kernel-argument segment and the common epilogue (bias, alpha, pad columns, atomics for
// split-K / accumulating problems, squared-error epilogue, zero-fill spans).
#pragma once
#include "internal.h"

namespace mfm {

#define MFM_GEMM_NEPI 4     // problems per launch that may carry an output transform (always the first ones)

struct GemmProblem {
  MfmGemmDesc d;
  int tiles_m, tiles_n, block_begin, k_per_split;
};
struct GemmGroup {
  GemmProblem p[MFM_GEMM_MAXP];
  int begins[MFM_GEMM_MAXP];     // first workgroup of every problem (INT_MAX past `count`): found with one unrolled compare chain
  int count;
  // optional: spans the launch also clears (the fused step's loss slots and gradient buffer ride on its
  // first GEMM instead of two memset launches of ~4.7 us each)
  float* zero_ptr[MFM_GEMM_ZSPANS];
  int64_t zero_n[MFM_GEMM_ZSPANS];
  // optional: squared-error epilogue for the first mse_count problems (decoder fc1 -> x_hat): the tile that
  // produced x_hat also forms d x_hat and its share of the reconstruction loss (mfm_mosi.py:441-446), so the
  // separate elementwise launch and its re-read of x_hat disappear
  MseEpi mse[3];
  int mse_count;
  // optional: per-problem output transforms (internal.h::GemmEpi), first epi_count problems
  GemmEpi epi[MFM_GEMM_NEPI];
  int epi_count, epi_train;
  unsigned long long epi_seed;
  const unsigned long long* epi_tick;
};


// Epilogue of one workgroup tile.  acc[fm][fn] follows the 16x16 MFMA accumulator map (row = 4*(lane>>4) + r,
// col = lane & 15), identical for the f32 and bf16 instructions.
template <int FR, int TF>
__device__ __forceinline__ void gemm_epilogue(const GemmGroup& g, const MfmGemmDesc& d, const int pi, const int z,
                                              const int split, const int m0, const int n0, const int wm, const int wn,
                                              const int bi, const int q, const int tid, const int lane, const int wave,
                                              f32x4 (&acc)[FR][FR], float (&tgt)[TF][TF][4], const bool do_mse) {
  const MseEpi& me = g.mse[do_mse ? pi : 0];
  float* __restrict__ C = d.c ? d.c + (int64_t)z * d.c_sz : nullptr;
  float* __restrict__ C2 = d.c2 ? d.c2 + (int64_t)z * d.c_sz : nullptr;
  // c_bf16 (bf16-resident outputs of a bf16 plan: the x-projection, dH): the same element offsets, 2-byte elements
  const bool c16 = d.c_bf16 != 0;
  __bf16* __restrict__ C16 = reinterpret_cast<__bf16*>(d.c) + (int64_t)z * d.c_sz;
  __bf16* __restrict__ C216 = reinterpret_cast<__bf16*>(d.c2) + (int64_t)z * d.c_sz;
  float sq = 0.0f;
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
        if (pi < g.epi_count && col < d.n_valid) {          // wave-uniform condition on pi
          const GemmEpi& ep = g.epi[pi];
          if (ep.kind == 1) {
            float mk = 1.0f;
            if (g.epi_train && ep.p > 0.0f) {
              const uint64_t idx = ((uint64_t)ep.op_id << 40) + (uint64_t)row * (uint64_t)d.n_valid + (uint64_t)col;
              mk = (rng_uniform(g.epi_seed + (g.epi_tick ? *g.epi_tick : 0ull), idx) < ep.p) ? 0.0f : 1.0f / (1.0f - ep.p);
            }
            ep.aux[off] = (v > 0.0f) ? mk : 0.0f;
            v = fmaxf(v, 0.0f) * mk;
          } else if (ep.kind == 2) {
            v = act_tanh(v);
          } else if (ep.kind == 3) {
            v *= ep.aux[off];
          }
        }
        if (d.accumulate) {
          if (col < d.n_valid) {
            atomicAdd(C + off, v);
            if (C2) atomicAdd(C2 + off, v);
          }
        } else {
          if (c16) {
            C16[off] = (__bf16)v;
            if (C2) C216[off] = (__bf16)v;
          } else {
            if (C) C[off] = v;          // (null: a training step needs the squared error and d x_hat only, not x_hat itself)
            if (C2) C2[off] = v;
          }
          if (do_mse && col < d.n_valid) {
            const float diff = v - tgt[fm % TF][fn % TF][r];
            sq += diff * diff;
            if (me.dxhat) {
              const int64_t doff = me.ld_dxhat ? (int64_t)row * me.ld_dxhat + col : off;
              if (me.dxhat_bf16) reinterpret_cast<__bf16*>(me.dxhat)[doff] = (__bf16)(me.grad_scale * diff);
              else me.dxhat[doff] = me.grad_scale * diff;
            }
          }
        }
      }
    }
  if (do_mse && me.loss) {
    __shared__ float sqsum[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    if (lane == 0) sqsum[wave] = sq;
    lds_barrier();
    if (tid == 0) atomicAdd(me.loss, (sqsum[0] + sqsum[1] + sqsum[2] + sqsum[3]) * me.inv_count);
  }
  // zero-fill spans last: ahead of the K loop these stores would sit in front of the first tile loads in the
  // in-order vmcnt queue and put a store round trip on every workgroup's critical path
#pragma unroll
  for (int zi = 0; zi < MFM_GEMM_ZSPANS; ++zi) {
    if (g.zero_n[zi] > 0) {          // 16-byte aligned, multiple of 4 floats (checked by the host)
      f32x4* z4 = reinterpret_cast<f32x4*>(g.zero_ptr[zi]);
      const int64_t n4 = g.zero_n[zi] >> 2;
      for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n4; i += (int64_t)gridDim.x * 256) z4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
}

// gemm_bf16.hip
int gemm_bf16_launch_kernel(const GemmGroup& g, int FR, int total, hipStream_t stream);
constexpr int BKB = 64;   // K depth of one LDS stage of the bf16 kernel: two 32-deep MFMA k-steps

}  // namespace mfm
