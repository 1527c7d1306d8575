// The decoder CHAIN of a training step at small batches (round 6): decoder recurrence -> fc1 + squared error + dH -> decoder BPTT
// in ONE launch, one workgroup per (decoder, batch row).
//
// Reference: decoderLSTM.forward incl. fc1 (mfm_model.py:64-91), the three reconstruction terms of the joint loss
// (mfm_mosi.py:433-439) and what loss.backward() sends back through fc1 and the decoder's time loop (mfm_mosi.py:440).
//
// Why.  The launch clock (profiles/r06_launch_timeline.txt) showed the three launches of this chain -- recurrence 22.8 us, fc1 +
// MSE + dH 10.4 us, BPTT 24.6 us -- plus three ~1.6 us launch gaps: 62 us of a 152 us step, on 96 / 240 / 96 workgroups.  Nothing
// in the chain crosses batch rows: x_hat[t, b] = Wfc h_t(b) + b, its error and dH[t, b] = dx_hat[t, b] Wfc belong to row b alone.
// So the workgroup that ran row b's recurrence turns its own T hidden states into dH itself -- a [T x h] x [h x d] product, its
// error, and a [T x d] x [d x h] product on the fp32 MFMA, Wfc from L2 -- and walks straight into its BPTT: no launch boundary,
// no grid-wide wait for the slowest row, no cold prologue of a third kernel, and dH needs neither atomics nor a zeroed buffer
// (the fc1 launch spread a row tile's columns over workgroups to fill the chip).  The weight gradients (dWfc, the decoders'
// dW) stay where they were: they read d x_hat, dA and the hidden states from memory later.
//
// This header: the fc1 phase of one row (dec_fc1_row_body).  lstm_seq_small.hip holds the kernel and its launcher.
#pragma once
#include "internal.h"
#include "lstamp.h"

namespace mfm {

constexpr int DCH_MAXJ = 8;        // Hp <= 128
constexpr int DCH_MAXRT = 4;       // T <= 64 (16-row MFMA tiles)
constexpr int DCH_MAXD = 320;      // output columns of one decoder
constexpr int DCH_OOB = 0x7FFFFFF0;          // buffer offset beyond any resource: the load returns 0

// LDS of the fc1 phase: hidden rows [R][Hp + 4], d x_hat [R][16 NF1 + 4] (both strides: an odd number of 16-byte groups), partial sums
static inline size_t dch_lds_floats(int T, int Hp, int d) {
  const size_t R = (size_t)((T + 15) / 16) * 16;
  return R * (size_t)(Hp + 4) + R * (size_t)((d + 15) / 16 * 16 + 4) + 32;
}

__device__ __forceinline__ float dch_rnd(float x, bool on) { return on ? (float)(__bf16)x : x; }

// One batch row b of one decoder: rows of the products are the T time steps.  Every wave of the workgroup takes part.
//   product 1: wave w owns the column fragments f = w, w + nw, ... (16 output columns each) for ALL row tiles: its Wfc rows are
//              requested once (16-byte loads along the hidden units, as in dec_fc1.hip) and meet every row tile from LDS;
//   epilogue : x_hat (optional), the squared error, d x_hat -> memory (the weight-gradient launch reads it) and -> LDS;
//   product 2: wave w owns the (row tile, 16 hidden units) fragments p = w, w + nw, ...; Wfc columns from L2 in two register
//              sets of 16 reduction steps that alternate between "requested" and "multiplied".
__device__ __forceinline__ void dec_fc1_row_body(const DecFc1Item& I, const int T, const int B, const int b, float* lds, const bool rb) {
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
  const int bi = lane & 15, q = lane >> 4;
  const int d = I.d, h = I.h, Hp = I.Hp;
  const int RT = (T + 15) >> 4, R = RT * 16;
  const int NF1 = (d + 15) >> 4, J = Hp >> 4;
  const int LDH = Hp + 4, LDD = NF1 * 16 + 4;
  float* Ht = lds;                       // [R][LDH]
  float* Dx = lds + R * LDH;             // [R][LDD]
  float* red = Dx + R * LDD;             // [nw]
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)I.w, 0, d * h * 4, 0x00020000);
  LSTAMP(2, 0);

  f32x4 w1[DCH_MAXJ];
  float xv[DCH_MAXRT][4], bv = 0.0f;
  auto load_f = [&](const int f) {
    const int n = f * 16 + bi;
    const bool cok = n < d;
#pragma unroll
    for (int j = 0; j < DCH_MAXJ; ++j) {
      const bool ok = (int)cok & (int)(j < J);
      // (units >= h of a weight row belong to the next row -- finite values, or zeros past the buffer -- and meet the exact
      //  zeros of the hidden tile's pad units)
      w1[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, ok ? (n * h + 16 * j + 4 * q) * 4 : DCH_OOB, 0, 0));
    }
#pragma unroll
    for (int rt = 0; rt < DCH_MAXRT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rt * 16 + 4 * q + r;
        xv[rt][r] = (cok && row < T) ? I.x[((int64_t)row * B + b) * I.ldx + n] : 0.0f;
      }
    bv = cok ? I.bias[n] : 0.0f;
  };
  // this wave's first fragment is requested before the hidden rows are parked (uniform branch)
  if (wave < NF1) load_f(wave);
  LSTAMP(2, 3);
  {
    const int per_row = Hp >> 2;
    for (int idx = tid; idx < R * per_row; idx += nt) {
      const int row = idx / per_row, k4 = idx - row * per_row;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row < T) v = *reinterpret_cast<const f32x4*>(I.hs + ((int64_t)row * B + b) * Hp + 4 * k4);
      if (rb) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = dch_rnd(v[e], true);
      }
      *reinterpret_cast<f32x4*>(Ht + row * LDH + 4 * k4) = v;
    }
  }
  LSTAMP(2, 4);
  lds_barrier();
  LSTAMP(2, 1);

  float lsum = 0.0f;
  for (int f = wave; f < NF1; f += nw) {
    if (f != wave) load_f(f);
    const int n = f * 16 + bi;
    const bool cok = n < d;
#if MFM_LAUNCH_STAMP
    if (f == wave) LSTAMP_W(2, 5);
#endif
#pragma unroll
    for (int rt = 0; rt < DCH_MAXRT; ++rt) if (rt < RT) {      // (no `break`: the loop must stay unrolled, xv[rt] in registers)
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < DCH_MAXJ; ++j) {
        const f32x4 hv = *reinterpret_cast<const f32x4*>(Ht + (rt * 16 + bi) * LDH + 16 * min(j, J - 1) + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = mma16x16x4(hv[e], dch_rnd(w1[j][e], rb), acc);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rt * 16 + 4 * q + r;
        float dx = 0.0f;
        if (cok && row < T) {
          const float xh = acc[r] + bv;
          const float diff = xh - xv[rt][r];
          lsum = fmaf(diff, diff, lsum);
          dx = I.grad_scale * diff;
          const int64_t o = ((int64_t)row * B + b) * d + n;
          if (I.xhat) I.xhat[o] = xh;
          if (I.dxhat) I.dxhat[o] = dx;
        }
        Dx[row * LDD + f * 16 + bi] = dch_rnd(dx, rb);
      }
    }
#if MFM_LAUNCH_STAMP
    if (f == wave) LSTAMP(2, 6);
#endif
  }
  LSTAMP(2, 7);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) lsum += __shfl_xor(lsum, o, 64);
  if (lane == 0) red[wave] = lsum;
  lds_barrier();
  LSTAMP(2, 2);
  if (tid == 0 && I.loss) {
    float s = 0.0f;
    for (int w = 0; w < nw; ++w) s += red[w];
    atomicAdd(I.loss, s * I.inv_count);
  }

  // ---- product 2: dH[R, Hp] = dx_hat[R, 16 NF1] Wfc[d, h]   (masked weights are zeros: pad units come out as exact zeros)
  const int KS = NF1 * 4;                 // 4-wide reduction steps
  for (int p = wave; p < RT * J; p += nw) {
    const int rt = p / J, cf = p - rt * J;
    const int c = cf * 16 + bi;
    const bool cvalid = c < h;
    auto load_w2 = [&](const int ks0, float (&wv)[16]) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = 4 * (ks0 + i) + q;
        const bool ok = (int)(ks0 + i < KS) & (int)(k < d) & (int)cvalid;
        wv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wres, ok ? (k * h + c) * 4 : DCH_OOB, 0, 0));
      }
    };
    f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
    const float* dxp = Dx + (rt * 16 + bi) * LDD + q;
    auto mult = [&](const int ks0, const float (&wv)[16]) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc2 = mma16x16x4(dxp[4 * min(ks0 + i, KS - 1)], dch_rnd(wv[i], rb), acc2);
    };
    float wa[16], wb[16];
    load_w2(0, wa);
#if MFM_LAUNCH_STAMP
    if (p == wave) LSTAMP_W(2, 8);
#endif
    for (int ks0 = 0; ks0 < KS; ks0 += 32) {
      load_w2(ks0 + 16, wb);
      mult(ks0, wa);
      load_w2(ks0 + 32, wa);
      mult(ks0 + 16, wb);
    }
#if MFM_LAUNCH_STAMP
    if (p == wave) LSTAMP(2, 9);
#endif
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rt * 16 + 4 * q + r;
      if (row < T) I.dhs[((int64_t)row * B + b) * Hp + c] = acc2[r];
    }
  }
  LSTAMP(2, 15);
}

}  // namespace mfm
