// Whole-sequence LSTM recurrence for SMALL batch tiles (latency regime), forward and BPTT.
//
// Why a second kernel family.  The MFMA kernels of lstm_seq.hip need 16 batch rows per
// workgroup (the N dimension of v_mfma_f32_16x16x4_f32), so the reference's B=32 minibatch gives
// two workgroups per LSTM: 14 of 256 CUs busy, ~6 us per time step (profiles/r01a).  The fp32
// MFMA and the fp32 VALU have the SAME peak on gfx950 (64 FLOP/clk/SIMD), so when CUs are idle
// nothing is lost by running the recurrent product on the VALU with a finer batch tile and 4x
// more workgroups.  One workgroup owns R=4 batch rows of one LSTM and up to 1024 threads
// (16 waves: four per SIMD hide the LDS latency; <=128 VGPRs each, ~60 of them resident weights):
//   * forward: thread (unit u, gate pair p, k-slice q of 4) keeps W[(2p+{0,1})*h+u][4j+q] in VGPRs
//     for all T steps; h_{t-1} sits in LDS as [k][4 rows], read as one broadcast ds_read_b128
//     per k (8 FMAs per LDS read); the four partial sums of a quad are all-reduced with two DPP
//     quad_perm adds; the two gate-pair lanes of (u, row q) swap their two pre-activations (DPP)
//     and both run the pointwise LSTM math, so c stays in registers.
//   * backward: thread (unit pair p, k-slice q of 16) keeps W^T for its two units; dA_t sits in
//     LDS as [gate][HKB][4 rows]; partial dh are all-reduced over 16 lanes (quad_perm x2,
//     row_half_mirror, row_mirror); lane q<8 owns (unit 2p+(q>>2), row q&3) for the gate-gradient
//     math.
//   * weights reach the registers through LDS: each gate's [h x h] block is copied coalesced into
//     a panel and every thread picks its (strided / transposed) elements from there -- per-thread
//     global gathers of W^T cost ~30-90 us per launch (profiles/r01 seq micro-benchmark).
// Buffers, layouts and the encoder/decoder forms are exactly those of lstm_seq.hip.
//
// This header: the device bodies of one workgroup's recurrence (small_fwd_body / small_bwd_body) and their helpers;
// lstm_seq_small.hip instantiates them in the kernels (plain, fold, role-workgroup forms) and holds the launchers.
#pragma once
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "internal.h"
#include "latent_row_dev.h"
#include "lstm_seq_dev.h"
#include "proj_role_dev.h"
#include "dw_role_dev.h"
#include "lstamp.h"

namespace mfm {

// Barrier that also waits for this wave's outstanding GLOBAL stores.  `__syncthreads()` alone does not: its workgroup-scope
// release needs no vmcnt wait on gfx950 outside threadgroup-split mode (the waves of a workgroup share one L1), so a flag
// raised right behind it can overtake the data it announces -- measured round 4: 32 wrong weight gradients in 6000 steps at
// T <= 2, always the smallest encoder, none with the wait (profiles/r04_handover_safety.txt).
__device__ __forceinline__ void sync_stores() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// Debug build only (make MFM_EXTRA_FLAGS=-DMFM_SEQ_STAMP=k, scripts/seq_step_timeline.sh): a clock on the links of the one-row
// forward step.  Every wave takes the shader clock at the top of each time step (P0) and at ONE further point k of the step --
// 1: h_{t-1} has arrived from LDS, 2: the recurrent FMAs are issued, 3: gates / c / h are computed, 4: the step's LDS writes
// are acknowledged (in front of the barrier), 5: the barrier released this wave, 6: the next step's top (the whole period) --
// and sums Pk - P0 over the steps t >= 1; at the end lane 0 of wave w of workgroup 0 leaves the average in cs[T-1][0][w]
// (the build is for timing, its last cell-state row is garbage).  One point per build: two s_memtime per step keep the
// perturbation at a few cycles (profiles/r05_seq_step_timeline.txt).
#ifndef MFM_SEQ_STAMP
#define MFM_SEQ_STAMP 0
#endif
// 1: a step's record (forward) / dA (backward) is written to HBM during the NEXT phase of the loop, its LDS read batched with the
// product's operand reads; 0 (default): read + store right behind the barrier (rounds 1-4).  Built on the timeline's "12 % of a
// step between the barrier and the next step's top" and MEASURED SLOWER (whole step 0.1664 vs 0.1605 ms): the step is bound by
// VALU issue between the arrival of h_{t-1} and the last wave's gate math; the window behind the barrier is idle issue time, and
// work moved out of it into the product costs what it issues.  profiles/r05_seq_step_timeline.txt
#ifndef MFM_SEQ_LATE_WRITEOUT
#define MFM_SEQ_LATE_WRITEOUT 0
#endif
__device__ __forceinline__ unsigned long long seq_clock() {
  unsigned long long t;
  asm volatile("s_memtime %0" : "=s"(t) :: "memory");
  return t;
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}
constexpr int DPP_QUAD_XOR1 = 0xB1;        // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;        // quad_perm:[2,3,0,1]
constexpr int DPP_QUAD_REV = 0x1B;         // quad_perm:[3,2,1,0]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;
// lane i <-> lane i^4 inside each group of 8: half-mirror (i -> 7-i) then quad reverse (i -> i^3)
__device__ __forceinline__ float dpp_xor4(float x) { return dpp_f<DPP_QUAD_REV>(dpp_f<DPP_ROW_HALF_MIRROR>(x)); }

// Pin a descriptor field in scalar registers.  A lane-dependent choice between two fields of the by-value
// launch descriptor otherwise becomes a load from a lane-dependent kernarg address, and the compiler
// then copies the whole descriptor to scratch.
// The result is typed as a global-memory pointer: behind the asm the compiler no longer sees that the
// value came from a kernel argument and would fall back to flat_load / flat_store.
typedef __attribute__((address_space(1))) float gfloat;
__device__ __forceinline__ gfloat* pin_s(const float* p) {
  asm volatile("" : "+s"(p));
  return (gfloat*)p;
}

// R consecutive floats from LDS (one ds_read_b32/b64/b128)
template <int R> struct RowVec { float v[R]; };
template <int R>
__device__ __forceinline__ RowVec<R> ld_rows(const float* p) {
  RowVec<R> o;
  if constexpr (R == 4) { const f32x4 t = *reinterpret_cast<const f32x4*>(p); o.v[0] = t[0]; o.v[1] = t[1]; o.v[2] = t[2]; o.v[3] = t[3]; }
  else if constexpr (R == 2) { typedef float f32x2 __attribute__((ext_vector_type(2))); const f32x2 t = *reinterpret_cast<const f32x2*>(p); o.v[0] = t[0]; o.v[1] = t[1]; }
  else { o.v[0] = p[0]; }
  return o;
}
// element `r` (lane-dependent, r < R) of a register array without dynamic indexing
template <int R>
__device__ __forceinline__ float sel_row(const float (&a)[R], int r) {
  if constexpr (R == 4) { const float lo = (r & 1) ? a[1] : a[0], hi = (r & 1) ? a[3] : a[2]; return (r & 2) ? hi : lo; }
  else if constexpr (R == 2) { return (r & 1) ? a[1] : a[0]; }
  else { return a[0]; }
}

// Copy one gate's [h x h] weight block into the LDS panel, coalesced, 4 loads in flight per thread.
// mode 0: W_hh   1: W_ih   2: W_ih + W_hh (decoder steps >= 1, mfm_model.py:85)
__device__ __forceinline__ void stage_gate(const SeqDev& d, int mode, int g, float* __restrict__ panel, int tid,
                                           int nt) {
  const int n = d.h * d.h;
  const float* __restrict__ a = (mode == 0 ? d.w_hh : d.w_ih) + (int64_t)g * n;
  const float* __restrict__ b2 = d.w_hh + (int64_t)g * n;
  if ((n & 3) == 0) {
    const int n4 = n >> 2;
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(b2);
    f32x4* p4 = reinterpret_cast<f32x4*>(panel);
    for (int base = tid; base < n4; base += 4 * nt) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = min(base + u * nt, n4 - 1);
        v[u] = a4[i];
        if (mode == 2) v[u] += b4[i];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (base + u * nt < n4) p4[base + u * nt] = v[u];
    }
  } else {
    for (int i = tid; i < n; i += nt) panel[i] = (mode == 2) ? a[i] + b2[i] : a[i];
  }
}

// Two gates per round for the one-row BPTT prologue: both gates' [h x h] blocks are requested before either is parked in
// LDS, so the prologue pays two L2 round trips instead of four (a workgroup is alone on its CU and its bandwidth is what
// it keeps in flight: 4 loads per thread and round trip were ~14 B/clk, ~2 us per gate; decoders stage W_ih + W_hh).
__device__ __forceinline__ void stage_gates2(const SeqDev& d, int mode, int g, float* __restrict__ panel, int tid, int nt) {
  const int n = d.h * d.h;
  if ((n & 3) != 0) {          // scalar fallback
    stage_gate(d, mode, g, panel, tid, nt);
    stage_gate(d, mode, g + 1, panel + n, tid, nt);
    return;
  }
  const int n4 = n >> 2;
  const f32x4* a4 = reinterpret_cast<const f32x4*>((mode == 0 ? d.w_hh : d.w_ih) + (int64_t)g * n);
  const f32x4* b4 = reinterpret_cast<const f32x4*>(d.w_hh + (int64_t)g * n);
  f32x4* p4 = reinterpret_cast<f32x4*>(panel);
  // gates g and g + 1 are adjacent in memory: one range of 2 n4 vectors
  for (int base = tid; base < 2 * n4; base += 8 * nt) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = a4[min(base + u * nt, 2 * n4 - 1)];
    if (mode == 2) {
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] += b4[min(base + u * nt, 2 * n4 - 1)];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (base + u * nt < 2 * n4) p4[base + u * nt] = v[u];
  }
}

// --------------------------------------------------------------------------------- forward
// Global traffic goes through LDS so that it is issued by full, coalesced waves: a vector-memory
// instruction occupies the CU's address unit for ~16 clocks per wave whether 2 or 64 lanes carry data,
// and with one (unit, gate pair, k-slice) thread layout every wave would issue its own 3 stores and
// 2 loads per step (80 instructions per step, ~0.5 us of a 1.4 us step; profiles/r01 seq experiments).
// Instead the owners drop (i, f, g, o, c, h) into an LDS record and 6*Hp*R/64 waves write it out after
// the step's barrier; the encoders' x-projections are fetched two steps ahead by 4*Hp*R/64 waves.
// FLG (encoders of the fold launch with projection role workgroups, proj_role_dev.h): the x-projections of a time step
// are produced inside this launch; the fetching waves check the step's block flags (requested one step earlier) before they
// request its values, and read them with agent-scope loads.
// BF (one-row tiles, bf16 plans below the batch size of the bf16 MFMA kernels; round 4): the recurrent product on
// v_dot2c_f32_bf16 -- W packed as bf16 pairs in half the registers, h_{t-1} exchanged through LDS as bf16 (what the bf16 MFMA
// kernels feed their matrix cores: same rounding points), fp32 accumulation, gate math, cell state and saved activations.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// Returns where the row's LAST hidden state lies in LDS (one-row fp32 tiles: [HKB] floats, pad units exact zeros; else null).
template <int KQ, int R, bool FLG = false, bool BF = false>
__device__ __forceinline__ const float* small_fwd_body(const SeqDev& d, const int T, const int B, const int tile, float* lds,
                                               const unsigned* flg = nullptr, const unsigned epoch = 0, const int ncb = 0,
                                               const HoCtl ctl = HoCtl{nullptr, nullptr, nullptr, 5000000ll, 1u}) {
  constexpr int HK = 4 * KQ;                    // padded hidden extent of the matvec
  constexpr int HKB = (HK + 15) / 16 * 16;      // == Hp
  constexpr int NTH = 8 * HKB;                  // threads this LSTM uses (blockDim may be larger)
  constexpr int NXL = (4 * R + 7) / 8;          // x-projection elements fetched per thread and step
  constexpr int NOS = (6 * R + 7) / 8;          // output elements written per thread and step
  const int tid = threadIdx.x, nt = blockDim.x;
  const int q = tid & 3, gp = (tid >> 2) & 1, u = tid >> 3;
  const int h = d.h, Hp = d.Hp;
  const bool dec = d.is_dec != 0;
  const bool uact = u < Hp;
  const int myrow = q & (R - 1);              // lanes q >= R duplicate row q % R (compute only, no stores)
  const bool rowner = q < R;
  const int b0 = tile * R;
  const int b = b0 + myrow;

  // R == 1: the quad lane q owns the CONTIGUOUS columns k = 16m + 4q + {0..3}, m < NM, so one ds_read_b128
  // feeds 4 FMAs per gate row (8 LDS reads per step instead of 30) and the weights arrive as 16-byte global
  // loads straight into registers (no LDS staging round in the prologue).  R > 1: k = 4j + q as before.
  constexpr int NM = HKB / 16;
  constexpr int NWR = (R == 1) ? 4 * NM : KQ;     // weights per gate row and thread
  constexpr int HX = (R == 1) ? HKB : HK;          // extent of one h buffer
  static_assert(!BF || R == 1, "bf16 dot products: one-row tiles only");
  float* hbuf = lds;                       // [2][HX][R]
  __bf16* hb16 = reinterpret_cast<__bf16*>(lds);   // BF: the same two buffers as bf16 (half the bytes)
  float* panel = lds + 2 * HKB * R;        // [2][h][h] weight staging (two gates at a time)
  float* obuf = panel;                     // [2][6][HKB][R] step outputs; aliases the panel (barriers below)
  float* xbuf = obuf + 2 * 6 * HKB * R;    // [2][4][HKB][R] x-projections of the next step (encoders)

  float w[2][NWR];
  bf16x2_t wp[2][BF ? NWR / 2 : 1];         // BF: the resident weights as bf16 pairs (k, k + 1)
  auto pack_w = [&]() {
    if constexpr (BF) {
#pragma unroll
      for (int gl = 0; gl < 2; ++gl)
#pragma unroll
        for (int j = 0; j < NWR / 2; ++j) wp[gl][j] = bf16x2_t{(__bf16)w[gl][2 * j], (__bf16)w[gl][2 * j + 1]};
    }
  };
  // round gl stages gates gl (for the p=0 lanes) and 2+gl (p=1 lanes) side by side, so every
  // lane picks its own gate with an address, not a predicate
  auto load_w = [&](int mode) {
    if constexpr (R == 1) {
      const float* fimg = mode == 2 ? d.wf_img : (mode == 1 ? d.wf1_img : nullptr);
      if (fimg) {          // this step's W_ih + W_hh (or W_ih), already in register order (lstm_seq_dev.h)
        const f32x4* img = reinterpret_cast<const f32x4*>(fimg) + (tid < NTH ? tid : 0);
#pragma unroll
        for (int gl = 0; gl < 2; ++gl)
#pragma unroll
          for (int m = 0; m < NM; ++m) {
            const f32x4 v = img[(gl * NM + m) * NTH];
#pragma unroll
            for (int i = 0; i < 4; ++i) w[gl][4 * m + i] = (tid < NTH) ? v[i] : 0.0f;
          }
        pack_w();
        return;
      }
      if ((h & 3) == 0) {
        const float* wa = (mode == 0) ? d.w_hh : d.w_ih;
#pragma unroll
        for (int gl = 0; gl < 2; ++gl) {
          const int64_t rowo = ((int64_t)(2 * gp + gl) * h + min(u, h - 1)) * h;
#pragma unroll
          for (int m = 0; m < NM; ++m) {
            const int k0 = 16 * m + 4 * q;
            const bool ok = (u < h) && (k0 < h);
            const int kc = min(k0, h - 4);
            f32x4 v = *reinterpret_cast<const f32x4*>(wa + rowo + kc);
            if (mode == 2) v += *reinterpret_cast<const f32x4*>(d.w_hh + rowo + kc);
#pragma unroll
            for (int i = 0; i < 4; ++i) w[gl][4 * m + i] = ok ? v[i] : 0.0f;
          }
        }
        pack_w();
        return;
      }
    }
    const int n = h * h;
    const int uc = min(u, h - 1);
#pragma unroll
    for (int gl = 0; gl < 2; ++gl) {
      stage_gate(d, mode, gl, panel, tid, nt);
      stage_gate(d, mode, 2 + gl, panel + n, tid, nt);
      __syncthreads();
      const float* src = panel + gp * n + uc * h;
#pragma unroll
      for (int j = 0; j < NWR; ++j) {
        const int k = (R == 1) ? 16 * (j >> 2) + 4 * q + (j & 3) : 4 * j + q;
        const float v = src[min(k, h - 1)];
        w[gl][j] = (u < h && k < h) ? v : 0.0f;
      }
      __syncthreads();
    }
    pack_w();
  };
  load_w(dec ? 1 : 0);
  LSTAMP_W(dec ? 1 : 0, 1);

  // decoder: constant bias per lane; encoder: x projection (bias folded in by the GEMM) via xbuf
  float gxb[2] = {0.f, 0.f};
  if (dec && u < h) {
#pragma unroll
    for (int gl = 0; gl < 2; ++gl) gxb[gl] = d.b_ih[(2 * gp + gl) * h + u] + d.b_hh[(2 * gp + gl) * h + u];
  }
  if (dec) {
    for (int idx = tid; idx < HX * R; idx += nt) {
      const int k = idx / R, br = b0 + (idx % R);
      const float v = (k < h && br < B) ? d.h_init[(int64_t)br * d.ld_init + k] : 0.0f;
      if constexpr (BF) hb16[idx] = (__bf16)v; else hbuf[idx] = v;
    }
  }

  const int64_t row4 = 4 * (int64_t)Hp;
  const int64_t gstep = (int64_t)B * row4, sstep = (int64_t)B * Hp;
  // x-projection fetch elements: e -> (gate g, row r, unit), unit fastest (coalesced)
  const gfloat* xp[NXL];
  int xl[NXL];
  bool xok[NXL];
  float xpf[NXL];
#pragma unroll
  for (int i = 0; i < NXL; ++i) {
    const int e = tid + i * NTH;
    xok[i] = !dec && tid < NTH && e < 4 * HKB * R;
    const int ec = xok[i] ? e : 0;
    const int sr = ec / HKB, unit = ec % HKB, r = sr % R, g = sr / R;
    xp[i] = (const gfloat*)d.gates + ((int64_t)min(b0 + r, B - 1) * 4 + g) * Hp + unit;
    xl[i] = (g * HKB + unit) * R + r;
    xpf[i] = 0.0f;
  }
  // FLG: only the waves that fetch projections look at flags (wave-uniform)
  const bool fetch_wave = FLG && !dec && (tid & ~63) < 4 * HKB * R && (tid & ~63) < NTH;
  unsigned fv = 0;
  if (!dec) {
    if constexpr (FLG) {
      if (fetch_wave) {
        const int t1 = (T > 1) ? 1 : 0;
        proj_flags_wait(flg, proj_flags_load(flg, ncb), epoch, ncb, ctl);
        proj_flags_wait(flg + t1 * PROJ_ROLE_FLAGS, proj_flags_load(flg + t1 * PROJ_ROLE_FLAGS, ncb), epoch, ncb, ctl);
        fv = proj_flags_load(flg + min(2, T - 1) * PROJ_ROLE_FLAGS, ncb);
      }
    }
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
      float v0;
      if constexpr (FLG) { v0 = ld_agent((const float*)xp[i]); xpf[i] = ld_agent((const float*)xp[i] + ((T > 1) ? gstep : 0)); }
      else { v0 = xp[i][0]; xpf[i] = xp[i][(T > 1) ? gstep : 0]; }
      if (xok[i]) xbuf[xl[i]] = v0;
    }
  }
  // output elements: e -> (slot s in i,f,g,o,c,h ; row r ; unit)
  gfloat* const p_gates = pin_s(d.gates);
  gfloat* const p_cs = pin_s(d.cs);
  gfloat* const p_hs = pin_s(d.hs);
  gfloat* op[NOS];
  int64_t ostr[NOS];
  int ol[NOS];
  bool ook[NOS];
#pragma unroll
  for (int i = 0; i < NOS; ++i) {
    const int e = tid + i * NTH;
    const bool in = tid < NTH && e < 6 * HKB * R;
    const int ec = in ? e : 0;
    const int sr = ec / HKB, unit = ec % HKB, r = sr % R, sl = sr / R;
    ook[i] = in && (b0 + r < B);
    const int br = min(b0 + r, B - 1);
    op[i] = sl < 4 ? p_gates + ((int64_t)br * 4 + sl) * Hp + unit : (sl == 4 ? p_cs : p_hs) + (int64_t)br * Hp + unit;
    ostr[i] = sl < 4 ? gstep : sstep;
    ol[i] = (sl * HKB + unit) * R + r;
  }
  const int ucl = min(u, HKB - 1);
  const int my_o = (2 * gp * HKB + ucl) * R + myrow;       // this lane's slots: 2gp, 2gp+1 and 4+gp
  const float sc0 = gp ? 2.0f : 1.0f;                      // gate 0 of the pair: sigmoid(i) / tanh(g)
  __syncthreads();
  LSTAMP(dec ? 1 : 0, 2);

  float c = 0.0f;
  int cur = 0;
  bool pend = false;          // the previous step's record waits in obuf (uniform)
#if MFM_SEQ_STAMP
  unsigned long long st_sum = 0, st_prev = 0;
#endif
  auto step = [&](const int t) {
    const int par = t & 1;
#if MFM_LAUNCH_STAMP
    if (t == (dec ? 2 : 1)) LSTAMP(dec ? 1 : 0, 5);
    if (t == T - 1) LSTAMP(dec ? 1 : 0, 6);
#endif
#if MFM_SEQ_STAMP
    const unsigned long long st0 = seq_clock();
    unsigned long long st1 = st0;
    if (MFM_SEQ_STAMP == 6 && t >= 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st_sum += st0 - st_prev; }
    st_prev = st0;
#endif
    float gx0 = gxb[0], gx1 = gxb[1];
    if (!dec) {
      const float* xb = xbuf + par * (4 * HKB * R) + my_o;
      gx0 = xb[0]; gx1 = xb[HKB * R];
    }
    // the record of step t - 1 (dropped into obuf in front of the last barrier) leaves for HBM DURING this step: its LDS read is
    // requested together with h_{t-1}, the stores go out behind the product (round 5: written out right behind the barrier, the
    // read's round trip sat between the barrier and the next step's h reads -- 12 % of a step, profiles/r05_seq_step_timeline.txt)
    float rprev[NOS];
#pragma unroll
    for (int i = 0; i < NOS; ++i) rprev[i] = 0.0f;
    if (MFM_SEQ_LATE_WRITEOUT && pend) {
#pragma unroll
      for (int i = 0; i < NOS; ++i) rprev[i] = obuf[(par ^ 1) * (6 * HKB * R) + ol[i]];
    }
    float acc[2][R];
#pragma unroll
    for (int gl = 0; gl < 2; ++gl)
#pragma unroll
      for (int r = 0; r < R; ++r) acc[gl][r] = 0.0f;
    if constexpr (R == 1) {
      if constexpr (BF) {
        if (dec || t > 0) {
          const bf16x2_t* hb = reinterpret_cast<const bf16x2_t*>(hb16 + cur * HX + 4 * q);
          bf16x2_t hv[NM][2];
#pragma unroll
          for (int m = 0; m < NM; ++m) { hv[m][0] = hb[8 * m]; hv[m][1] = hb[8 * m + 1]; }      // one ds_read_b64 per m
#pragma unroll
          for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              acc[0][0] = __builtin_amdgcn_fdot2_f32_bf16(wp[0][2 * m + i], hv[m][i], acc[0][0], false);
              acc[1][0] = __builtin_amdgcn_fdot2_f32_bf16(wp[1][2 * m + i], hv[m][i], acc[1][0], false);
            }
        }
      } else if (dec || t > 0) {
        const float* hb = hbuf + cur * HX + 4 * q;
        f32x4 hv[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) hv[m] = *reinterpret_cast<const f32x4*>(hb + 16 * m);
#if MFM_SEQ_STAMP == 1
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        st1 = seq_clock();
#endif
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[0][0] = fmaf(w[0][4 * m + i], hv[m][i], acc[0][0]);
            acc[1][0] = fmaf(w[1][4 * m + i], hv[m][i], acc[1][0]);
          }
#if MFM_SEQ_STAMP == 2
        asm volatile("s_memtime %0" : "=s"(st1), "+v"(acc[0][0]), "+v"(acc[1][0]) :: "memory");
#endif
      }
    } else if (dec || t > 0) {
      const float* hb = hbuf + cur * (HK * R) + q * R;
      constexpr int RING = (KQ < 4) ? KQ : 4;      // LDS reads kept in flight ahead of their FMAs
      RowVec<R> ring[RING];
#pragma unroll
      for (int j = 0; j < RING; ++j) ring[j] = ld_rows<R>(hb + 4 * R * j);
#pragma unroll
      for (int j = 0; j < KQ; ++j) {
        const RowVec<R> hv = ring[j % RING];                               // the R rows of k = 4j+q
        if (j + RING < KQ) ring[j % RING] = ld_rows<R>(hb + 4 * R * (j + RING));
#pragma unroll
        for (int gl = 0; gl < 2; ++gl)
#pragma unroll
          for (int r = 0; r < R; ++r) acc[gl][r] = fmaf(w[gl][j], hv.v[r], acc[gl][r]);
      }
    }
    if (MFM_SEQ_LATE_WRITEOUT && pend) {
#pragma unroll
      for (int i = 0; i < NOS; ++i) {
        if (ook[i]) *op[i] = rprev[i];
        op[i] += ostr[i];
      }
    }
    pend = true;
    // x-projections of step t+2 (clamped re-read at the tail), issued well ahead of their LDS hand-over
    float xn[NXL];
#pragma unroll
    for (int i = 0; i < NXL; ++i) xn[i] = 0.0f;
    if (!dec) {
      const int64_t off = (int64_t)min(t + 2, T - 1) * gstep;
      if constexpr (FLG) {
        if (fetch_wave) {
          proj_flags_wait(flg + min(t + 2, T - 1) * PROJ_ROLE_FLAGS, fv, epoch, ncb, ctl);
#pragma unroll
          for (int i = 0; i < NXL; ++i) xn[i] = ld_agent((const float*)xp[i] + off);
          fv = proj_flags_load(flg + min(t + 3, T - 1) * PROJ_ROLE_FLAGS, ncb);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NXL; ++i) xn[i] = xp[i][off];
      }
    }
    // all-reduce the four k-slices of the quad, keep the sums of batch row q, add bias / x-projection
    float mine[2];
#pragma unroll
    for (int gl = 0; gl < 2; ++gl) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float v = acc[gl][r];
        v += dpp_f<DPP_QUAD_XOR1>(v);
        v += dpp_f<DPP_QUAD_XOR2>(v);
        acc[gl][r] = v;
      }
      mine[gl] = sel_row<R>(acc[gl], myrow) + (gl ? gx1 : gx0);
    }
    // each lane activates its own two gates; the partner lane (same unit and row, other gate pair)
    // supplies the other two.  i*g is symmetric in the exchange, so both lanes carry c.
    const float a0 = act_scaled(mine[0], sc0);     // gp 0: i   gp 1: g
    const float a1 = act_sigmoid(mine[1]);         // gp 0: f   gp 1: o
    const float p0 = dpp_xor4(a0), p1 = dpp_xor4(a1);
    const float gf = gp ? p1 : a1, go = gp ? a1 : p1;
    c = fmaf(gf, c, a0 * p0);          // explicit: every instantiation must round the same way
    float hv = go * act_tanh(c);
#if MFM_SEQ_STAMP == 3
    asm volatile("s_memtime %0" : "=s"(st1), "+v"(hv), "+v"(c) :: "memory");
#endif
    if (uact && rowner) {
      float* ob = obuf + par * (6 * HKB * R) + my_o;
      ob[0] = a0; ob[HKB * R] = a1;
      ob[(4 - gp) * HKB * R] = gp ? hv : c;        // slot 4 (c) from gp 0, slot 5 (h) from gp 1
      if (gp == 0 && u < HX) {
        const float hn = (b < B) ? hv : 0.0f;
        if constexpr (BF) hb16[(cur ^ 1) * HX + u] = (__bf16)hn; else hbuf[(cur ^ 1) * (HX * R) + u * R + myrow] = hn;
      }
    }
    if (!dec) {
#pragma unroll
      for (int i = 0; i < NXL; ++i)
        if (xok[i]) xbuf[(par ^ 1) * (4 * HKB * R) + xl[i]] = xpf[i];
    }
#if MFM_SEQ_STAMP == 4
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    st1 = seq_clock();
#endif
    lds_barrier();
#if MFM_SEQ_STAMP == 5
    st1 = seq_clock();
#endif
#if MFM_SEQ_STAMP
    if (MFM_SEQ_STAMP != 6 && t >= 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st_sum += st1 - st0; }
#endif
    if (!MFM_SEQ_LATE_WRITEOUT) {
#pragma unroll
      for (int i = 0; i < NOS; ++i) {
        if (ook[i]) *op[i] = obuf[par * (6 * HKB * R) + ol[i]];
        op[i] += ostr[i];
      }
      pend = false;
    }
#pragma unroll
    for (int i = 0; i < NXL; ++i) xpf[i] = xn[i];
    cur ^= 1;
  };
  // the record of the last step taken (still in obuf) -> HBM
  auto flush = [&](const int t_last) {
    if (!pend) return;
#pragma unroll
    for (int i = 0; i < NOS; ++i) {
      if (ook[i]) *op[i] = obuf[(t_last & 1) * (6 * HKB * R) + ol[i]];
      op[i] += ostr[i];
    }
    pend = false;
  };
  // The decoder's step 0 (W_ih on the embedding) is peeled so that the weight reload sits between
  // two clean loops instead of inside one (keeps its temporaries out of the hot loop's registers).
  if (dec) {
    step(0);
    flush(0);                          // (the record of step 0 lives in the panel about to be refilled)
    LSTAMP(1, 3);
    if (T > 1) {
      __syncthreads();
      load_w(2);                       // steps >= 1 feed h back as the input: W_ih + W_hh
      LSTAMP_W(1, 4);
      for (int t = 1; t < T; ++t) step(t);
      flush(T - 1);
    }
  } else {
    for (int t = 0; t < T; ++t) step(t);
    flush(T - 1);
  }
  LSTAMP(dec ? 1 : 0, 7);
#if MFM_SEQ_STAMP
  __syncthreads();
  if (blockIdx.x == 0 && (tid & 63) == 0 && T > 2) {
    const int nsteps = (MFM_SEQ_STAMP == 6) ? T - 2 : T - 1;
    p_cs[(int64_t)(T - 1) * sstep + (tid >> 6)] = (float)((double)st_sum / (double)nsteps);
  }
#endif
  if constexpr (R == 1 && !BF) return hbuf + cur * HX;
  else return nullptr;
}

// --------------------------------------------------------------------------------- backward
// Same LDS hand-over as the forward: the saved activations of step t-2 are fetched by 7*Hp*R/64 full waves
// during step t and published one step later; dA_t leaves through the dA panel the matvec reads anyway.
// KS = k-slices per unit pair (lanes that share a pair of output units and all-reduce their partial sums).  16: the layout
// described at the top (8 * Hp threads).  8 (one-row tiles only): half the threads with twice the FMAs each and contiguous
// gate columns per lane (c = 32 m + 4 q + e, one ds_read_b128 per 4 columns).  Built on the hypothesis that the step is
// VALU-issue bound (~165 instructions per wave and step, 56 of them the recurrent FMAs); measured no faster (see
// seq_small_launch), so it is opt-in (MFM_SEQ_KS=8) and parity-tested only.
// PUB (encoders of the fold launch with weight-gradient role workgroups, dw_role_dev.h): dA_t leaves with agent-scope stores
// and one step later, once those stores are acknowledged, stamp[t * DWR_ROWS] <- epoch tells the role workgroups that this
// row's dA of the time steps >= t is in memory.
// LAT = LatentDev (fold launch with the latent chain in front, round 6): once the transposed weights are REQUESTED and before
// anything of them is used, the chain's stores and atomics that nothing in this workgroup waits for are issued (latent_bwd_row_body,
// mode 2) -- they go out while the weights travel; `dh_lds`: the encoder's d h_T in LDS (the chain's gradient record) instead of
// memory; `late_stamp`: raised behind the prologue's barrier once every store of the workgroup so far has been acknowledged (the
// chain's gradient record is readable by the weight-gradient role workgroups).  (Plain reference parameters on purpose: a lambda
// that captured the by-value LatentDev kernel argument made the compiler copy the descriptor to scratch, 1.1 KB per lane.)
template <int KQ, int R, int KS, bool PUB, class LAT>
__device__ __forceinline__ void small_bwd_body_x(const SeqDev& d, const int T, const int B, const int tile, float* lds,
                                                 unsigned* stamp, const unsigned epoch, const bool skip_final_stamp,
                                                 const LAT& lat, const float* lat_params, float* lat_grads, const int lat_ch,
                                                 const float* dh_lds, unsigned* late_stamp) {
  constexpr int HK = 4 * KQ;                       // padded hidden extent
  constexpr int HKB = (HK + 15) / 16 * 16;         // per-gate extent of the dA panel (multiple of 16) == Hp
  static_assert(KS == 16 || (KS == 8 && R == 1), "8 k-slices: one-row tiles only");
  constexpr int NG = HKB / 16;                     // gate columns per thread and gate (KS == 16)
  constexpr int NW = 4 * HKB / KS;                 // gate columns per thread: KS 16: k = g*HKB + 16 i + q;  KS 8: c = 32 m + 4 q + e
  constexpr int NB = NW / 4;                       // 16-byte column blocks per thread (KS == 8)
  constexpr int NTH = (KS / 2) * HKB;
  constexpr int NV = 7;                            // staged values per (unit,row): gi gf gg go c_{t-1} dh_ext dc_ext
  constexpr int NLD = (NV * R + KS / 2 - 1) / (KS / 2);     // fetch elements per thread and step
  constexpr int NST = (4 * R + KS / 2 - 1) / (KS / 2);      // dA elements written per thread and step
  const int tid = threadIdx.x, nt = blockDim.x;
  const int q = tid & (KS - 1), up = tid / KS;
  const int h = d.h, Hp = d.Hp;
  const bool dec = d.is_dec != 0;
  const bool has_dc = d.dc_ext != nullptr;
  const int ua = 2 * up, ub = 2 * up + 1;           // the two output units whose W^T rows this thread holds
  const int mu = 2 * up + ((q >> 2) & 1), mr = q & 3;   // the (unit,row) lanes q<8 own in the pointwise part
  const bool own = (q < 8) && (mr < R) && (mu < Hp);
  const int b0 = tile * R;
  const int b = b0 + mr;
  const bool bvalid = own && (b < B);

  float* dabuf = lds;                       // [2][4][HKB][R]
  float* sbuf = lds + 2 * 4 * HKB * R;      // [2][NV][HKB][R] saved activations of the coming step
  float* panel = sbuf + 2 * NV * HKB * R;   // [1 or 2][h][h] weight staging (one-row tiles: two gates per round)

  float wa[NW], wb[NW];
  auto load_wT = [&](int mode) {
    // this step's transposed weights, already in thread order (written by the role workgroups of the forward launch,
    // proj_role_dev.h): coalesced loads straight into the registers instead of two staging rounds through LDS
    if constexpr (KS == 16 && R == 1) {
      if (d.wt_img && mode != 1) {
        // (threads beyond NTH own no unit and share no reduction group with one that does: what they hold is never used)
        static_assert(NW % 4 == 0, "transposed-weight image: quads of slots");
        const f32x4* img = reinterpret_cast<const f32x4*>(d.wt_img) + (tid < NTH ? tid : 0);
#pragma unroll
        for (int i = 0; i < NW / 4; ++i) {
          const f32x4 va = img[(int64_t)i * NTH], vb = img[(int64_t)(NW / 4 + i) * NTH];
#pragma unroll
          for (int e = 0; e < 4; ++e) { wa[4 * i + e] = va[e]; wb[4 * i + e] = vb[e]; }
        }
        return;
      }
    }
    const int uac = min(ua, h - 1), ubc = min(ub, h - 1);
    constexpr int PG = (R == 1) ? 2 : 1;          // gates staged per round (the panel holds PG blocks: small_lds_bytes)
#pragma unroll
    for (int g0 = 0; g0 < 4; g0 += PG) {
      if constexpr (PG == 2) stage_gates2(d, mode, g0, panel, tid, nt);
      else stage_gate(d, mode, g0, panel, tid, nt);
      __syncthreads();
#pragma unroll
      for (int gg = 0; gg < PG; ++gg) {
        const int g = g0 + gg;
        const float* pan = panel + gg * h * h;
        if constexpr (KS == 16) {
#pragma unroll
          for (int i = 0; i < NG; ++i) {
            const int j = 16 * i + q;                    // unit index of this gate column
            const int jc = min(j, h - 1);
            const float va = pan[jc * h + uac], vb = pan[jc * h + ubc];
            wa[g * NG + i] = (j < h && ua < h) ? va : 0.0f;
            wb[g * NG + i] = (j < h && ub < h) ? vb : 0.0f;
          }
        } else {
          // block m of this lane = flat gate columns 32 m + 4 q + {0..3}; a block never straddles two gates (HKB % 4 == 0)
#pragma unroll
          for (int m = 0; m < NB; ++m) {
            const int c0 = 32 * m + 4 * q;
            const int cg = c0 / HKB, j0 = c0 - cg * HKB;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int j = j0 + e;
              const int jc = min(j, h - 1);
              const float va = pan[jc * h + uac], vb = pan[jc * h + ubc];
              const float na = (j < h && ua < h) ? va : 0.0f, nb = (j < h && ub < h) ? vb : 0.0f;
              if (g == 0) { wa[4 * m + e] = 0.0f; wb[4 * m + e] = 0.0f; }
              wa[4 * m + e] = (cg == g) ? na : wa[4 * m + e];
              wb[4 * m + e] = (cg == g) ? nb : wb[4 * m + e];
            }
          }
        }
      }
      __syncthreads();
    }
  };
  load_wT(dec ? 2 : 0);
  if constexpr (std::is_same<LAT, LatentDev>::value) latent_bwd_row_body<false>(lat, lat_params, lat_grads, tile, lat_ch, lds, 2);
  LSTAMP_W(dec ? 3 : 4, 1);

  const int64_t row4 = 4 * (int64_t)Hp;
  const int64_t gstep = (int64_t)B * row4, sstep = (int64_t)B * Hp;
  const int nv = has_dc ? 7 : (dec ? 6 : 5);       // slots actually fetched
  // fetch elements: e -> (slot v, row r, unit), unit fastest.  Every fetch is unconditional (inactive
  // elements re-read element 0): a load under a branch makes the compiler's in-order vmcnt bookkeeping
  // conservative and the wait for the older prefetch would also cover the younger one.
  gfloat* const p_gates = pin_s(d.gates);
  const gfloat* const p_cs = pin_s(d.cs);
  const gfloat* const p_dh = pin_s(d.dh_ext);
  const gfloat* const p_dc = pin_s(d.dc_ext);
  const gfloat* fp[NLD];
  int64_t fstr[NLD];
  int fl[NLD];
  bool fok[NLD], fcp[NLD];
  float pf[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int e = tid + i * NTH;
    fok[i] = tid < NTH && e < nv * HKB * R;
    const int ec = fok[i] ? e : 0;
    const int sr = ec / HKB, unit = ec % HKB, r = sr % R, v = sr / R;
    const int br = min(b0 + r, B - 1);
    fcp[i] = (v == 4);
    const gfloat* src = (v == 5 && dec) ? p_dh : (v == 6 && has_dc) ? p_dc : p_cs;
    fp[i] = v < 4 ? p_gates + ((int64_t)br * 4 + v) * Hp + unit : src + (int64_t)br * Hp + unit;
    fstr[i] = v < 4 ? gstep : sstep;
    fl[i] = (v * HKB + unit) * R + r;
    if (v == 5 && !dec) fok[i] = false;            // encoders take dL/dh_T only (below)
  }
  // element i at step tt: slot 4 holds c_{tt-1} (zero at tt == 0), everything else is indexed by tt itself
  auto fetch = [&](int i, int tt) {
    const int ti = fcp[i] ? max(tt - 1, 0) : tt;
    const float v = fp[i][(int64_t)ti * fstr[i]];
    return (fcp[i] && tt == 0) ? 0.0f : v;
  };
  // dA write-out elements: e -> (row r, gate g, unit): one contiguous 4*Hp run per row
  gfloat* sp[NST];
  int sl[NST];
  bool sok[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int e = tid + i * NTH;
    const bool in = tid < NTH && e < 4 * HKB * R;
    const int ec = in ? e : 0;
    const int r = ec / (4 * HKB), rem = ec % (4 * HKB);
    sok[i] = in && (b0 + r < B);
    sp[i] = p_gates + ((int64_t)(T - 1) * B + min(b0 + r, B - 1)) * row4 + rem;
    sl[i] = rem * R + r;
  }
  // pipeline prologue: step T-1 goes straight to LDS, step T-2 waits in registers
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const float v0 = fetch(i, T - 1);
    pf[i] = fetch(i, max(T - 2, 0));
    if (fok[i]) sbuf[((T - 1) & 1) * (NV * HKB * R) + fl[i]] = v0;
  }
  const int muc = min(mu, HKB - 1), mrc = mr & (R - 1);
  const int my_s = muc * R + mrc;
  float ct = d.cs[((int64_t)(T - 1) * B + min(b, B - 1)) * Hp + muc];
  float ext0 = 0.0f;
  if (!dec && mu < h) ext0 = dh_lds ? dh_lds[mu] : d.dh_ext[(int64_t)min(b, B - 1) * d.ld_dh + mu];   // dL/dh_T only
  float dh_rec = 0.0f, dc = 0.0f;
  int cur = 0;
  if (late_stamp) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (uniform) every store of this wave so far has been acknowledged
  __syncthreads();
  if (late_stamp && tid == 0) dwr_stamp(late_stamp, epoch);
  LSTAMP(dec ? 3 : 4, 2);

  auto step = [&](const int t, const bool matvec = true) {
    const int par = t & 1;
#if MFM_LAUNCH_STAMP
    if (t == T - 2) LSTAMP(dec ? 3 : 4, 5);
    if (t == 1) LSTAMP(dec ? 3 : 4, 6);
    if (t == 0) LSTAMP(dec ? 3 : 4, 3);
#endif
    const float* sb = sbuf + par * (NV * HKB * R) + my_s;
    const float gi = sb[0], gf = sb[HKB * R], gg = sb[2 * HKB * R], go = sb[3 * HKB * R], cp = sb[4 * HKB * R];
    float ext = ext0, dce = 0.0f;
    if (dec) ext = sb[5 * HKB * R];
    if (has_dc) dce = sb[6 * HKB * R];
    ext0 = 0.0f;
    // (the peeled decoder step 0 -- matvec == false, a constant of that call site -- prefetches nothing: its fetch would queue
    // behind the W_ih rows requested in front of it and hold the step's own stores back)
    float pn[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) pn[i] = matvec ? fetch(i, max(t - 2, 0)) : 0.0f;
    const float dh = dh_rec + ext;
    const float tc = act_tanh(ct);
    const float dot = dh * tc;
    const float dct = dh * go * (1.0f - tc * tc) + dc + dce;
    float da[4];
    da[0] = dct * gg * gi * (1.0f - gi);
    da[1] = dct * cp * gf * (1.0f - gf);
    da[2] = dct * gi * (1.0f - gg * gg);
    da[3] = dot * go * (1.0f - go);
    dc = dct * gf;
    ct = cp;                            // c_{t-1} is the cell state of the next (earlier) step
    if (!bvalid) { da[0] = 0.0f; da[1] = 0.0f; da[2] = 0.0f; da[3] = 0.0f; }
    float* db = dabuf + cur * (4 * HKB * R);
    if (own) {
#pragma unroll
      for (int g = 0; g < 4; ++g) db[(g * HKB + muc) * R + mr] = da[g];
    }
    if (t > 0) {
#pragma unroll
      for (int i = 0; i < NLD; ++i)
        if (fok[i]) sbuf[(par ^ 1) * (NV * HKB * R) + fl[i]] = pf[i];
    }
    // PUB: everything older than this step's NLD prefetch loads -- the dA stores of step t + 1 among it -- has been
    // acknowledged (asked for as late as the step allows: the write-through stores have had a whole step)
    if constexpr (PUB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
    lds_barrier();
    if constexpr (PUB) {
      if (tid == 0 && t + 1 < T) dwr_stamp(stamp + (t + 1) * DWR_ROWS, epoch);
    }
    // dA_t leaves for HBM from registers BEHIND the product: its LDS read is requested together with the product's operands
    // (round 5: read + store right behind the barrier put one LDS round trip in front of the product's reads)
    float wr[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) wr[i] = db[sl[i]];
    auto write_da = [&]() {
#pragma unroll
      for (int i = 0; i < NST; ++i) {
        if constexpr (PUB) { if (sok[i]) __hip_atomic_store((float*)sp[i], wr[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else { if (sok[i]) *sp[i] = wr[i]; }
        sp[i] -= gstep;
      }
    };
    if (!MFM_SEQ_LATE_WRITEOUT) write_da();
#pragma unroll
    for (int i = 0; i < NLD; ++i) pf[i] = pn[i];
    if (matvec && ((t > 0) || dec)) {
      float aa[R], ab[R];
#pragma unroll
      for (int r = 0; r < R; ++r) { aa[r] = 0.0f; ab[r] = 0.0f; }
      if constexpr (KS == 16) {
        const float* dp = db + q * R;
        constexpr int RING = (NW < 4) ? NW : 4;
        RowVec<R> ring[RING];
#pragma unroll
        for (int i = 0; i < RING; ++i) ring[i] = ld_rows<R>(dp + 16 * R * i);
#pragma unroll
        for (int i = 0; i < NW; ++i) {
          const RowVec<R> dv = ring[i % RING];                               // the R rows of column 16i+q
          if (i + RING < NW) ring[i % RING] = ld_rows<R>(dp + 16 * R * (i + RING));
#pragma unroll
          for (int r = 0; r < R; ++r) {
            aa[r] = fmaf(wa[i], dv.v[r], aa[r]);
            ab[r] = fmaf(wb[i], dv.v[r], ab[r]);
          }
        }
      } else {
        const float* dp = db + 4 * q;
        constexpr int RING = (NB < 4) ? NB : 4;
        f32x4 ring[RING];
#pragma unroll
        for (int i = 0; i < RING; ++i) ring[i] = *reinterpret_cast<const f32x4*>(dp + 32 * i);
#pragma unroll
        for (int m = 0; m < NB; ++m) {
          const f32x4 dv = ring[m % RING];                                   // gate columns 32 m + 4 q + {0..3}
          if (m + RING < NB) ring[m % RING] = *reinterpret_cast<const f32x4*>(dp + 32 * (m + RING));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            aa[0] = fmaf(wa[4 * m + e], dv[e], aa[0]);
            ab[0] = fmaf(wb[4 * m + e], dv[e], ab[0]);
          }
        }
      }
      // all-reduce over the k-slices (after the two quad steps every lane of a quad holds the quad's sum, so the mirror
      // steps only have to bring in the other quads)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float v = aa[r];
        v += dpp_f<DPP_QUAD_XOR1>(v); v += dpp_f<DPP_QUAD_XOR2>(v);
        v += dpp_f<DPP_ROW_HALF_MIRROR>(v);
        if constexpr (KS == 16) v += dpp_f<DPP_ROW_MIRROR>(v);
        aa[r] = v;
        v = ab[r];
        v += dpp_f<DPP_QUAD_XOR1>(v); v += dpp_f<DPP_QUAD_XOR2>(v);
        v += dpp_f<DPP_ROW_HALF_MIRROR>(v);
        if constexpr (KS == 16) v += dpp_f<DPP_ROW_MIRROR>(v);
        ab[r] = v;
      }
      const float sa = sel_row<R>(aa, mrc), sb2 = sel_row<R>(ab, mrc);
      dh_rec = (q & 4) ? sb2 : sa;
    }
    if (MFM_SEQ_LATE_WRITEOUT) write_da();
    cur ^= 1;
  };
  for (int t = T - 1; t >= 1; --t) step(t);
  if (dec && (h & 3) == 0 && R == 1) {
    // The decoder's step-0 input gradient goes through W_ih alone: d h_init = dA_0 W_ih.  Re-loading the
    // whole transposed register layout for this one product costs ~7 us (four staging rounds); instead the
    // 4h rows of W_ih are streamed once with 16-byte loads (thread = 4 consecutive units x one row slice)
    // and the slices are summed through LDS.
    // (round 6, the launch clock: 4 us between the top of step 0 and the end of this product.)  The W_ih rows do not depend on
    // step 0: they are requested BEFORE its pointwise part -- the resident transposed weights are dead by now, their registers
    // take the rows -- and multiplied behind it.
    const int dcur = cur;
    const int h4 = h >> 2;
    const int S = min(min(32, nt / h4), h);         // row slices (S*h partial sums must fit the h*h panel)
    const int u4 = tid % h4, sl = tid / h4;
    constexpr int NPRE = (4 * HKB + 31) / 32;       // rows per thread at S = 32 (fewer slices: the loop below takes the rest)
    // (round 6, the launch clock again: 4.2 us from the requests to the last wave's partial sums, and the same with half of the
    // rows parked in LDS beforehand or with every line of W_ih touched in the prologue -- not memory: the loop divided every row
    // index by h to find its (gate, unit), ~30 VALU instructions x 13 rows x 13 waves on 4 SIMDs.)  Rows advance by S <= h:
    // the (gate, unit) pair and the row pointer advance by additions.
    f32x4 wpre[NPRE];
    const float* const wp0 = d.w_ih + (int64_t)min(sl, 4 * h - 1) * h + 4 * min(u4, h4 - 1);
    const int64_t wstep = (int64_t)S * h;
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const float* wp = (sl + i * S < 4 * h) ? wp0 + i * wstep : wp0;
      wpre[i] = *reinterpret_cast<const f32x4*>(wp);
    }
    LSTAMP(3, 8);
    step(0, false);
    LSTAMP(3, 9);
    const float* db = dabuf + dcur * (4 * HKB * R);
    if (sl < S) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      int dj = sl, dg = 0;                            // row sl + i S = gate dg, unit dj  (sl < S <= h)
#pragma unroll
      for (int i = 0; i < NPRE; ++i) {
        const float a = (sl + i * S < 4 * h) ? db[min(dg, 3) * HKB + dj] : 0.0f;
        acc += a * wpre[i];
        dj += S;
        if (dj >= h) { dj -= h; ++dg; }
      }
      for (int r = sl + NPRE * S; r < 4 * h; r += S) {
        const int gg = r / h, j = r - gg * h;
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(d.w_ih + (int64_t)r * h + 4 * u4);
        acc += db[gg * HKB + j] * w4;
      }
      float* pr = panel + sl * h + 4 * u4;
      pr[0] = acc[0]; pr[1] = acc[1]; pr[2] = acc[2]; pr[3] = acc[3];
    }
    LSTAMP(3, 10);
    lds_barrier();
    LSTAMP(3, 11);
    if (h <= 128 && nt >= 1024 && S == 32) {
      // 8 groups of 128 threads sum 4 slices each (one batch of LDS reads), the first h threads the 8 group sums
      const int grp = tid >> 7, u = tid & 127;
      float* const p2 = panel + S * h;                          // (S + 8) h <= 2 h^2: h >= 32 here (S == 32 <= h)
      if (u < h) {
        const float* pp = panel + (4 * grp) * h + u;
        p2[grp * h + u] = (pp[0] + pp[h]) + (pp[2 * h] + pp[3 * h]);
      }
      lds_barrier();
      if (tid < h && d.d_h_init && b0 < B) {
        float s8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) s8[k] = p2[k * h + tid];
        d.d_h_init[(int64_t)b0 * d.ld_dinit + tid] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
      }
    } else if (tid < h && d.d_h_init && b0 < B) {
      float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
      int k = 0;
      for (; k + 4 <= S; k += 4) {
        s0 += panel[k * h + tid]; s1 += panel[(k + 1) * h + tid]; s2 += panel[(k + 2) * h + tid]; s3 += panel[(k + 3) * h + tid];
      }
      for (; k < S; ++k) s0 += panel[k * h + tid];
      d.d_h_init[(int64_t)b0 * d.ld_dinit + tid] = (s0 + s1) + (s2 + s3);
    }
    LSTAMP(3, 7);
    return;
  }
  if (dec) load_wT(1);                 // grad wrt the step-0 input goes through W_ih only (peeled)
  step(0);
  if (dec && bvalid && d.d_h_init && mu < h) d.d_h_init[(int64_t)b * d.ld_dinit + mu] = dh_rec;
  if constexpr (PUB) {
    sync_stores();                     // every dA store of every wave has been acknowledged
    if (tid == 0 && !skip_final_stamp) dwr_stamp(stamp, epoch);      // (skipped only by the fault injection of the tests)
  }
  LSTAMP(dec ? 3 : 4, 7);
}

template <int KQ, int R, int KS = 16, bool PUB = false>
__device__ __forceinline__ void small_bwd_body(const SeqDev& d, const int T, const int B, const int tile, float* lds,
                                               unsigned* stamp = nullptr, const unsigned epoch = 0,
                                               const bool skip_final_stamp = false) {
  small_bwd_body_x<KQ, R, KS, PUB, int>(d, T, B, tile, lds, stamp, epoch, skip_final_stamp, 0, nullptr, nullptr, 0, nullptr, nullptr);
}

}  // namespace mfm
