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
  // optional: optimizer tail (internal.h::AdamTail).  Workgroups [0, gemm_blocks) are tiles, the rest apply Adam.
  AdamTail adam;
  int gemm_blocks;
};

// ---- optimizer tail.  Workgroups of one launch are dispatched in block order, so when a tail workgroup starts every tile
// workgroup is already resident or finished: the wait below cannot starve them (no deadlock by construction).  A tile
// workgroup publishes with an agent-scope release (its stores / atomics acknowledged, its XCD's L2 written back) before it
// counts itself in; a tail workgroup requests p, m, v first (nobody else touches them in this launch), polls the counter,
// then reads the gradients with agent-scope loads.
// accumulate = the tile's outputs were device-scope atomic adds (every weight-gradient product): they are performed at the
// device's coherence point, so waiting for their acknowledgement is enough; plain stores sit in this XCD's L2 and need the
// agent-scope release (an L2 write-back: ~1 us each and serialised per XCD -- 1000 tile workgroups doing that made the
// launch 185 us instead of 25).
// Arrivals are counted in two levels: 64 slot counters on separate 128-byte lines (agent-scope atomics on ONE address are
// serialised at the memory side, ~0.1 us each: a thousand arrivals on a single counter made the tail wait 100 us), and the
// workgroup that completes a slot counts it into the master word the tail polls.
__device__ __forceinline__ void gemm_tail_signal(const GemmGroup& g, const bool accumulate) {
  if (g.adam.counter) {
    if (accumulate) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
      const int slot = blockIdx.x & (MFM_TAIL_SLOTS - 1);
      const int expect = g.gemm_blocks / MFM_TAIL_SLOTS + (slot < g.gemm_blocks % MFM_TAIL_SLOTS ? 1 : 0);
      const int before = __hip_atomic_fetch_add(g.adam.counter + (1 + slot) * MFM_TAIL_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (before + 1 == expect) __hip_atomic_fetch_add(g.adam.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

constexpr int ADAM_TAIL_EPT = 4;      // float4 elements per tail thread
__device__ __forceinline__ void gemm_adam_tail(const GemmGroup& g) {
  const AdamTail& A = g.adam;
  const int64_t n4 = A.n >> 2;
  const int64_t base = (int64_t)((int)blockIdx.x - g.gemm_blocks) * (blockDim.x * ADAM_TAIL_EPT) + threadIdx.x;
  f32x4 pv[ADAM_TAIL_EPT], mv[ADAM_TAIL_EPT], vv[ADAM_TAIL_EPT];
#pragma unroll
  for (int e = 0; e < ADAM_TAIL_EPT; ++e) {
    const int64_t i = base + (int64_t)e * blockDim.x;
    const int64_t ic = i < n4 ? i : n4 - 1;
    pv[e] = reinterpret_cast<const f32x4*>(A.p)[ic];
    mv[e] = reinterpret_cast<const f32x4*>(A.m)[ic];
    vv[e] = reinterpret_cast<const f32x4*>(A.v)[ic];
  }
  if (threadIdx.x == 0) {
    const int slots = g.gemm_blocks < MFM_TAIL_SLOTS ? g.gemm_blocks : MFM_TAIL_SLOTS;     // slots that see an arrival
    const int64_t t0 = wall_clock64();
    while (__hip_atomic_load(A.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < slots) {
      __builtin_amdgcn_s_sleep(16);                     // ~0.4 us between polls
      if (wall_clock64() - t0 > 20000000ll) break;      // 0.2 s at 100 MHz: a bug, not a wait -- never hang the device
    }
  }
  __syncthreads();
  // the gradients are read with agent-scope loads (past this XCD's L2) instead of an acquire fence: the fence is an L2
  // invalidate per wave, serialised per XCD -- 1900 of them took ~115 us
  typedef unsigned long long u64;
  u64 lo[ADAM_TAIL_EPT], hi[ADAM_TAIL_EPT];
#pragma unroll
  for (int e = 0; e < ADAM_TAIL_EPT; ++e) {
    const int64_t i = base + (int64_t)e * blockDim.x;
    const int64_t ic = i < n4 ? i : n4 - 1;
    const u64* gp = reinterpret_cast<const u64*>(A.g + 4 * ic);
    lo[e] = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    hi[e] = __hip_atomic_load(gp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int e = 0; e < ADAM_TAIL_EPT; ++e) {
    const int64_t i = base + (int64_t)e * blockDim.x;
    if (i >= n4) continue;
    float a[2], b[2];
    __builtin_memcpy(a, &lo[e], 8);
    __builtin_memcpy(b, &hi[e], 8);
    const f32x4 gv = f32x4{a[0], a[1], b[0], b[1]};
    f32x4 p4 = pv[e], m4 = mv[e], v4 = vv[e];
#pragma unroll
    for (int j = 0; j < 4; ++j) {       // same arithmetic, in the same order, as adam_kernel (elementwise.hip)
      const float gg = gv[j] * A.grad_scale;
      m4[j] = m4[j] + (1.0f - A.beta1) * (gg - m4[j]);
      v4[j] = v4[j] * A.beta2 + (1.0f - A.beta2) * gg * gg;
      const float denom = sqrtf(v4[j]) / A.bc2_sqrt + A.eps;
      p4[j] = p4[j] - A.step_size * m4[j] / denom;
    }
    reinterpret_cast<f32x4*>(A.p)[i] = p4;
    reinterpret_cast<f32x4*>(A.m)[i] = m4;
    reinterpret_cast<f32x4*>(A.v)[i] = v4;
  }
}

// Epilogue of one workgroup tile.  acc[fm][fn] follows the 16x16 MFMA accumulator map (row = 4*(lane>>4) + r,
// col = lane & 15), identical for the f32 and bf16 instructions.
template <int FR, int TF>
__device__ __forceinline__ void gemm_epilogue(const GemmGroup& g, const MfmGemmDesc& d, const int pi, const int z,
                                              const int split, const int m0, const int n0, const int wm, const int wn,
                                              const int bi, const int q, const int tid, const int lane, const int wave,
                                              f32x4 (&acc)[FR][FR], float (&tgt)[TF][TF][4], const bool do_mse) {
  const MseEpi& me = g.mse[do_mse ? pi : 0];
  float* __restrict__ C = d.c + (int64_t)z * d.c_sz;
  float* __restrict__ C2 = d.c2 ? d.c2 + (int64_t)z * d.c_sz : nullptr;
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
              mk = (rng_uniform(g.epi_seed, idx) < ep.p) ? 0.0f : 1.0f / (1.0f - ep.p);
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
          C[off] = v;
          if (C2) C2[off] = v;
          if (do_mse && col < d.n_valid) {
            const float diff = v - tgt[fm % TF][fn % TF][r];
            sq += diff * diff;
            if (me.dxhat) me.dxhat[off] = me.grad_scale * diff;
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
      for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n4; i += (int64_t)g.gemm_blocks * 256) z4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
}

// gemm_bf16.hip
int gemm_bf16_launch_kernel(const GemmGroup& g, int FR, int total, hipStream_t stream);
constexpr int BKB = 64;   // K depth of one LDS stage of the bf16 kernel: two 32-deep MFMA k-steps

}  // namespace mfm
