// Whole-sequence LSTM recurrence, forward and BPTT, for gfx950.
//
// Design (MI355X-first, not a cell-at-a-time port of nn.LSTMCell):
//   * batch rows are independent through the recurrence, so the unit of parallelism is a
//     16-row batch tile: one persistent workgroup per (LSTM, tile) walks all T steps.  Several
//     LSTMs (the 4 encoders, or the 3 decoders) share ONE launch; nothing is exchanged between
//     workgroups, so there is no grid barrier and no inter-CU traffic.
//   * wave w of the workgroup owns hidden units [16w, 16w+16) for all four gates.  Its slice
//     of the recurrent weight matrix lives in VGPRs as MFMA A-operands for the whole kernel
//     (fwd: W[g*h+u][k], 4*ceil(h/4) registers; bwd: W^T, same count) -- weights are read from
//     HBM/L2 exactly once per launch, never per step.
//   * gates^T[4h x 16] = W[4h x h] * h_{t-1}^T[h x 16] on v_mfma_f32_16x16x4_f32.  With this
//     orientation the accumulator of lane l holds gates i,f,g,o of units 16w+4(l>>4)+{0..3}
//     for batch row l&15: the LSTM pointwise math is lane-local and the cell state c stays in
//     registers across all steps.  Only h_t crosses waves, through a double-buffered LDS
//     panel laid out [unit][16 rows] so the B-operand read is lds[64*kk + lane] (linear,
//     conflict-free) and one barrier per step suffices.
//   * backward mirrors it: dA_t (pre-activation gate grads) are lane-local, exchanged through
//     LDS as [gate*HK+unit][16 rows], and dh_{t-1}^T = W^T * dA_t^T runs on the same MFMA with
//     four independent accumulator chains.  dA overwrites the saved gates in place; the weight
//     gradients are batched GEMMs over dA afterwards (plan.hip).
#include <algorithm>
#include <type_traits>

#include <stdlib.h>

#include "internal.h"
#include "lstm_seq_dev.h"
#include "proj_role_dev.h"
#include "dw_role_dev.h"

namespace mfm {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// --------------------------------------------------------------------------------- forward
template <int HK4, int KIND>
__device__ __forceinline__ void seq_fwd_body(const SeqDev& d, const int T, const int B, const int tile,
                                             float* lds) {
  constexpr int HK = HK4 * 4;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int bi = lane & 15, q = lane >> 4;
  const int h = d.h, Hp = d.Hp;
  const bool active = wave < (Hp >> 4);
  const int u0 = wave * 16;
  const int b = tile * 16 + bi;
  const bool bvalid = active && (b < B);
  // KIND 0: encoder-only kernel (no weight-swap code at all); KIND 1: decoder kernel -- the flag is still read
  // from the descriptor there: with `dec` a compile-time constant hipcc hoists so much of the weight-swap code
  // that the one-off prologue spills ~300 registers, which a 20-step kernel does notice
  const bool dec = (KIND != 0) && (d.is_dec != 0);

  float w[4][HK4];
  // Branch-free weight fetch (pad elements read element 0 and are multiplied by 0).
  // MODE 0: W_hh (encoder)   1: W_ih (decoder step 0)   2: W_ih + W_hh (decoder steps >= 1)
  auto load_w = [&](auto mode, int zofs) {
    constexpr int MODE = decltype(mode)::value;
    const int unit = u0 + bi;
    const int uc = min(unit, h - 1);
    const int uok = (int)active & (int)(unit < h);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int kk = 0; kk < HK4; ++kk) {
        const int k = 4 * kk + q;
        const int off = (g * h + uc) * h + min(k, h - 1) + zofs;   // always a valid address
        const float m = (float)(uok & (int)(k < h));            // 0 for pad elements
        if constexpr (MODE == 0) w[g][kk] = d.w_hh[off] * m;
        else if constexpr (MODE == 1) w[g][kk] = d.w_ih[off] * m;
        else if constexpr (MODE == 2) w[g][kk] = (d.w_ih[off] + d.w_hh[off]) * m;
        else w[g][kk] += d.w_hh[off] * m;     // MODE 3: W_ih is resident, add W_hh (same single rounding as MODE 2)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (dec) load_w(std::integral_constant<int, 1>{}, 0); else load_w(std::integral_constant<int, 0>{}, 0);

  f32x4 bias[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    bias[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (dec && active) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int unit = u0 + 4 * q + r;
        if (unit < h) bias[g][r] = d.b_ih[g * h + unit] + d.b_hh[g * h + unit];
      }
    }
  }

  float* hbuf = lds;  // [2][HK*16]
  if (dec) {
    for (int idx = tid; idx < HK * 16; idx += blockDim.x) {
      const int unit = idx >> 4, br = tile * 16 + (idx & 15);
      hbuf[idx] = (unit < h && br < B) ? d.h_init[(int64_t)br * d.ld_init + unit] : 0.0f;
    }
    __syncthreads();
  }

  const int64_t row4 = 4 * (int64_t)Hp;
  f32x4 gx[4];
  if (!dec) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
      gx[g] = bvalid ? ld4(d.gates + ((int64_t)b) * row4 + g * Hp + u0 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
  }

  float c[4] = {0.f, 0.f, 0.f, 0.f};
  int cur = 0;
  // One time step.  The decoder swaps its weights once (W_ih for step 0, W_ih + W_hh afterwards,
  // mfm_model.py:83-85); that reload sits BETWEEN two calls of the step, not inside the time loop: a
  // reload in the loop makes every weight register loop-variant and the h >= 104 instantiations then
  // spill their weights and re-read them from scratch on every step (51-168 spilled registers).
  auto step = [&](const int t) {
    f32x4 acc[4];
    const int64_t rowt = (int64_t)t * B + b;
    if (dec) {
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = bias[g];
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = gx[g];
      if (t + 1 < T && bvalid) {
#pragma unroll
        for (int g = 0; g < 4; ++g) gx[g] = ld4(d.gates + (rowt + B) * row4 + g * Hp + u0 + 4 * q);
      }
    }
    if (active && (dec || t > 0)) {
      const float* hb = hbuf + cur * (HK * 16) + lane;
#pragma unroll
      for (int kk = 0; kk < HK4; ++kk) {
        const float hv = hb[64 * kk];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = mma16x16x4(w[g][kk], hv, acc[g]);
      }
    }
    if (active) {
      f32x4 gi, gf, gg, go, cv, hv;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        gi[r] = act_sigmoid(acc[0][r]);
        gf[r] = act_sigmoid(acc[1][r]);
        gg[r] = act_tanh(acc[2][r]);
        go[r] = act_sigmoid(acc[3][r]);
        c[r] = gf[r] * c[r] + gi[r] * gg[r];
        cv[r] = c[r];
        hv[r] = go[r] * act_tanh(c[r]);
      }
      if (bvalid) {
        float* gp = d.gates + rowt * row4 + u0 + 4 * q;
        st4(gp, gi); st4(gp + Hp, gf); st4(gp + 2 * Hp, gg); st4(gp + 3 * Hp, go);
        st4(d.cs + rowt * Hp + u0 + 4 * q, cv);
        st4(d.hs + rowt * Hp + u0 + 4 * q, hv);
      }
      float* hn = hbuf + (cur ^ 1) * (HK * 16);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int unit = u0 + 4 * q + r;
        if (unit < HK) hn[unit * 16 + bi] = (b < B) ? hv[r] : 0.0f;
      }
    }
    lds_barrier();
    cur ^= 1;
  };
  int t0 = 0;
  if (KIND != 0 && dec) {
    step(0);
    if (T > 1) load_w(std::integral_constant<int, 3>{}, 0);   // W_ih (step 0) -> W_ih + W_hh (steps >= 1)
    t0 = 1;
  }
  for (int t = t0; t < T; ++t) step(t);
}

// --------------------------------------------------------------------------------- backward
template <int HK4, int KIND>
__device__ __forceinline__ void seq_bwd_body(const SeqDev& d, const int T, const int B, const int tile,
                                             float* lds) {
  constexpr int HK = HK4 * 4;   // padded hidden extent; reduction runs over 4*HK gate columns
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int bi = lane & 15, q = lane >> 4;
  const int h = d.h, Hp = d.Hp;
  const bool active = wave < (Hp >> 4);
  const int u0 = wave * 16;
  const int b = tile * 16 + bi;
  const bool bvalid = active && (b < B);
  // KIND 0: encoder-only kernel (no weight-swap code at all); KIND 1: decoder kernel -- the flag is still read
  // from the descriptor there: with `dec` a compile-time constant hipcc hoists so much of the weight-swap code
  // that the one-off prologue spills ~300 registers, which a 20-step kernel does notice
  const bool dec = (KIND != 0) && (d.is_dec != 0);

  float wT[HK];
  auto load_wT = [&](auto mode, int zofs) {
    constexpr int MODE = decltype(mode)::value;
    const int unit = u0 + bi;   // A row = output unit of dh
    const int uc = min(unit, h - 1);
    const int uok = (int)active & (int)(unit < h);
#pragma unroll
    for (int kk = 0; kk < HK; ++kk) {
      const int k = 4 * kk + q;           // gate column in the [4][HK] padded numbering
      const int g = k / HK, up = k % HK;
      const int off = (g * h + min(up, h - 1)) * h + uc + zofs;   // always a valid address
      const float m = (float)(uok & (int)(up < h));
      float v;
      if constexpr (MODE == 0) v = d.w_hh[off];
      else if constexpr (MODE == 1) v = d.w_ih[off];
      else v = d.w_ih[off] + d.w_hh[off];
      wT[kk] = v * m;
      if ((kk & 15) == 15) __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (dec) load_wT(std::integral_constant<int, 2>{}, 0); else load_wT(std::integral_constant<int, 0>{}, 0);

  float* dabuf = lds;  // [2][4*HK*16]
  const int64_t row4 = 4 * (int64_t)Hp;
  f32x4 dh_rec = f32x4{0.f, 0.f, 0.f, 0.f};
  float dc[4] = {0.f, 0.f, 0.f, 0.f};
  int cur = 0;
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};

  // One BPTT step; `first` marks t == 0, where the decoder's input gradient goes through W_ih only: that
  // weight swap lives in the peeled last call, outside the time loop (see the forward body).
  auto step = [&](const int t, auto first) {
    const int64_t rowt = (int64_t)t * B + b;
    f32x4 dh = dh_rec;
    if (bvalid) {
      if (dec) {
        const f32x4 e = ld4(d.dh_ext + rowt * Hp + u0 + 4 * q);
        dh += e;
      } else if (t == T - 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int unit = u0 + 4 * q + r;
          if (unit < h) dh[r] += d.dh_ext[(int64_t)b * d.ld_dh + unit];
        }
      }
    }
    f32x4 gi = zero4, gf = zero4, gg = zero4, go = zero4, ct = zero4, cp = zero4;
    float* gp = d.gates + rowt * row4 + u0 + 4 * q;
    if (bvalid) {
      gi = ld4(gp); gf = ld4(gp + Hp); gg = ld4(gp + 2 * Hp); go = ld4(gp + 3 * Hp);
      ct = ld4(d.cs + rowt * Hp + u0 + 4 * q);
      if (t > 0) cp = ld4(d.cs + (rowt - B) * Hp + u0 + 4 * q);
    }
    f32x4 dce = zero4;
    if (bvalid && d.dc_ext) dce = ld4(d.dc_ext + rowt * Hp + u0 + 4 * q);
    f32x4 dai, daf, dag, dao;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float tc = act_tanh(ct[r]);
      const float dot = dh[r] * tc;
      const float dct = dh[r] * go[r] * (1.0f - tc * tc) + dc[r] + dce[r];
      dai[r] = dct * gg[r] * gi[r] * (1.0f - gi[r]);
      daf[r] = dct * cp[r] * gf[r] * (1.0f - gf[r]);
      dag[r] = dct * gi[r] * (1.0f - gg[r] * gg[r]);
      dao[r] = dot * go[r] * (1.0f - go[r]);
      dc[r] = dct * gf[r];
    }
    if (bvalid) { st4(gp, dai); st4(gp + Hp, daf); st4(gp + 2 * Hp, dag); st4(gp + 3 * Hp, dao); }

    const bool need_rec = (t > 0) || dec;
    if (need_rec) {
      float* db = dabuf + cur * (4 * HK * 16);
      if (active) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int up = u0 + 4 * q + r;
          if (up < HK) {
            db[(0 * HK + up) * 16 + bi] = dai[r];
            db[(1 * HK + up) * 16 + bi] = daf[r];
            db[(2 * HK + up) * 16 + bi] = dag[r];
            db[(3 * HK + up) * 16 + bi] = dao[r];
          }
        }
      }
      lds_barrier();
      if constexpr (decltype(first)::value) {
        if (dec) load_wT(std::integral_constant<int, 1>{}, 0);   // grad wrt the step-0 input goes through W_ih only
      }
      f32x4 a0 = zero4, a1 = zero4, a2 = zero4, a3 = zero4;
      if (active) {
        const float* dp = db + lane;
#pragma unroll
        for (int kk = 0; kk < HK; kk += 8) {
          // 8 LDS reads, 8 MFMAs, then a scheduling fence: keeps the compiler from hoisting all
          // HK operand reads above the MFMA chain (which spills the resident weights).
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = dp[64 * (kk + j)];
          a0 = mma16x16x4(wT[kk + 0], v[0], a0);
          a1 = mma16x16x4(wT[kk + 1], v[1], a1);
          a2 = mma16x16x4(wT[kk + 2], v[2], a2);
          a3 = mma16x16x4(wT[kk + 3], v[3], a3);
          a0 = mma16x16x4(wT[kk + 4], v[4], a0);
          a1 = mma16x16x4(wT[kk + 5], v[5], a1);
          a2 = mma16x16x4(wT[kk + 6], v[6], a2);
          a3 = mma16x16x4(wT[kk + 7], v[7], a3);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      dh_rec = (a0 + a1) + (a2 + a3);
      cur ^= 1;
    }
  };
  if constexpr (KIND != 0) {
    for (int t = T - 1; t >= 1; --t) step(t, std::false_type{});
    step(0, std::true_type{});
  } else {
    for (int t = T - 1; t >= 0; --t) step(t, std::false_type{});
  }
  if (dec && bvalid && d.d_h_init) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int unit = u0 + 4 * q + r;
      if (unit < h) d.d_h_init[(int64_t)b * d.ld_dinit + unit] = dh_rec[r];
    }
  }
}

#define MFM_SEQ_CASES(BODY, KIND)                                                            \
  switch (d.hk4) {                                                                           \
    case 2: BODY<2, KIND>(d, L.T, L.B, tile, lds); break;                                                        \
    case 4: BODY<4, KIND>(d, L.T, L.B, tile, lds); break;                                                        \
    case 6: BODY<6, KIND>(d, L.T, L.B, tile, lds); break;                                                        \
    case 8: BODY<8, KIND>(d, L.T, L.B, tile, lds); break;                                                        \
    case 10: BODY<10, KIND>(d, L.T, L.B, tile, lds); break;                                                      \
    case 12: BODY<12, KIND>(d, L.T, L.B, tile, lds); break;                                                      \
    case 14: BODY<14, KIND>(d, L.T, L.B, tile, lds); break;                                                      \
    case 16: BODY<16, KIND>(d, L.T, L.B, tile, lds); break;                                                      \
    case 18: BODY<18, KIND>(d, L.T, L.B, tile, lds); break;                                                      \
    case 20: BODY<20, KIND>(d, L.T, L.B, tile, lds); break;                                                      \
    case 22: BODY<22, KIND>(d, L.T, L.B, tile, lds); break;                                                      \
    case 24: BODY<24, KIND>(d, L.T, L.B, tile, lds); break;                                                      \
    case 26: BODY<26, KIND>(d, L.T, L.B, tile, lds); break;                                                      \
    case 28: BODY<28, KIND>(d, L.T, L.B, tile, lds); break;                                                      \
    case 30: BODY<30, KIND>(d, L.T, L.B, tile, lds); break;                                                      \
    case 32: BODY<32, KIND>(d, L.T, L.B, tile, lds); break;                                                      \
    default: break;                                                                          \
  }

// One kernel per direction AND per kind (encoders / decoders): the encoder kernels then carry no weight-swap
// code and keep all weights of the h = 120 instantiation in registers (0 spills; 165-750 before).
template <bool BWD, int KIND>
__global__ __launch_bounds__(512) void lstm_seq_kernel(const SeqLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int di = 0;
  const int bid = blockIdx.x;
#pragma unroll 1
  for (int i = 1; i < L.count; ++i)
    if (bid >= L.d[i].block_begin) di = i;
  const SeqDev& d = L.d[di];
  const int tile = bid - d.block_begin;
  if (BWD) { MFM_SEQ_CASES(seq_bwd_body, KIND) } else { MFM_SEQ_CASES(seq_fwd_body, KIND) }
}

// Path choice.  The VALU kernels (lstm_seq_small.hip, 4 rows per workgroup) win while the chip
// is under-filled; the MFMA kernels (16 rows per workgroup) are the throughput path.  Override with
// MFM_SEQ_PATH=mfma|small (tests run both).
static bool use_small_path(int B) {
  const char* e = opt_get("MFM_SEQ_PATH");
  if (e && e[0] == 'm') return false;
  if (e && e[0] == 's') return true;
  return B <= 512;
}

// bf16 MFMA recurrences (lstm_seq_bf16.hip) work on 16-row batch tiles: 128 x 16 cells of gate math per workgroup and
// step (~1.4 us per step at h = 120 since the round-5 coalescing, 2.1 before), whatever B is.  Below ~128 rows the chip is
// better used by the fp32 VALU kernels' one-row workgroups (0.8 us per step, and exact fp32 products), so a bf16 plan keeps
// its recurrences on those until B reaches MFM_BF16_SEQ_MINB (default 128, with bf16-resident activations from the same
// size, plan_build.hip; round 2-4: 192).  Measured, MOSI sizes, ms per step, MFMA + resident vs one-row fp32 kernels:
// B = 112: 0.2735 vs 0.2631, 128: 0.2749 vs 0.2785, 144: 0.2786 vs 0.3247, 160: 0.2803 vs 0.3419, 176: 0.2838 vs 0.3612.
// The bf16 entry points of the ABI always run the bf16 kernels.
bool bf16_seq_pays(int B) {
  const char* e = opt_get("MFM_BF16_SEQ_MINB");
  return B >= (e ? atoi(e) : 128);
}

struct FoldArgs { const LatentDev* lat; const float* params; float* grads; ProjRole* pr; DwRole* dr; const float* const* wt_imgs;
                  const WtImgItem* img_items; int n_img_items; bool* img_written; const float* const* wf_imgs; int dec_tail; };

static int seq_launch(const MfmSeqDesc* descs_in, int count_in, int T, int B, bool bwd, hipStream_t stream, bool bf16 = false,
                      const FoldArgs* fold = nullptr) {
  MFM_REQUIRE(descs_in && count_in >= 1 && count_in <= MFM_MAX_SEQ, "lstm_seq: count %d out of range", count_in);
  MFM_REQUIRE(T >= 1 && B >= 1, "lstm_seq: T=%d B=%d", T, B);
  // LSTMs too wide for the weight-resident kernels take the step-by-step path (lstm_step.hip); the rest of
  // the group still shares one launch
  MfmSeqDesc descs[MFM_MAX_SEQ], wide[MFM_MAX_SEQ];
  int count = 0, nwide = 0;
  for (int i = 0; i < count_in; ++i) {
    const MfmSeqDesc& s = descs_in[i];
    MFM_REQUIRE(s.h >= 1, "lstm_seq[%d]: h=%d", i, s.h);
    MFM_REQUIRE(s.gates && s.hs && s.cs && s.w_hh, "lstm_seq[%d]: null buffer", i);
    if (s.is_dec) MFM_REQUIRE(s.w_ih && s.b_ih && s.b_hh && s.h_init, "lstm_seq[%d]: decoder needs w_ih/b/h_init", i);
    if (bwd) MFM_REQUIRE(s.dh_ext, "lstm_seq_bwd[%d]: dh_ext is null", i);
    const bool force = opt_get("MFM_SEQ_STEPWISE") != nullptr;          // testing: every LSTM step by step
    if (s.h > MFM_SEQ_MAX_RESIDENT_H || force) {
      MFM_REQUIRE(!s.store_bf16 && !s.h_last, "lstm_seq[%d]: h = %d takes the step-by-step fp32 path, which has no bf16-resident form", i, s.h);
      wide[nwide++] = s;
    } else descs[count++] = s;
  }
  if (fold && fold->dec_tail) {
    if (nwide || bf16 || !use_small_path(B) || opt_get("MFM_SEQ_KS") || opt_get("MFM_SEQ_ROWS")) return MFM_ERR_UNSUPPORTED;
  } else if (fold && !fold->lat && !fold->pr && !fold->dr) {      // (images only: a plain launch)
    if (fold->img_written) *fold->img_written = false;
    if (nwide || bf16 || !use_small_path(B) || opt_get("MFM_SEQ_KS") || opt_get("MFM_SEQ_ROWS")) fold = nullptr;
  } else if (fold && (nwide || bf16 || !use_small_path(B))) return MFM_ERR_UNSUPPORTED;
  if (nwide) {
    int rc = seq_stepwise(wide, nwide, T, B, bwd, stream);
    if (rc != MFM_OK) return rc;
  }
  if (count == 0) return MFM_OK;
  // Longest job first: workgroups are dispatched in block order, and once a launch needs more than one
  // round of workgroups (B >= 128 with four LSTMs) the widest LSTM must not start in the last round
  // (B=2048: encoder recurrences 154 -> 132 us forward, 163 -> 151 us backward).  While everything is
  // resident at once the caller's order is kept: widest-first measured 0.8 % slower per step at B=32.
  bool sorted = false;
  if ((long)count * B > (long)device_cus() && !opt_get("MFM_SEQ_KEEP_ORDER")) {
    std::stable_sort(descs, descs + count, [](const MfmSeqDesc& a, const MfmSeqDesc& b) { return a.h > b.h; });
    sorted = true;
  }
  SeqLaunch L;
  memset(&L, 0, sizeof(L));
  L.count = count; L.T = T; L.B = B;
  const int tiles = cdiv(B, 16);
  int total = 0, max_waves = 1;
  size_t lds_bytes = 0;
  for (int i = 0; i < count; ++i) {
    const MfmSeqDesc& s = descs[i];
    SeqDev& d = L.d[i];
    d.gates = s.gates; d.hs = s.hs; d.cs = s.cs;
    d.w_hh = s.w_hh; d.w_ih = s.w_ih; d.b_ih = s.b_ih; d.b_hh = s.b_hh;
    d.h_init = s.h_init; d.ld_init = s.ld_init;
    d.dh_ext = s.dh_ext; d.ld_dh = s.ld_dh;
    d.d_h_init = s.d_h_init; d.ld_dinit = s.ld_dinit;
    d.dc_ext = s.dc_ext;
    d.w_pack = bf16 ? s.w_pack : nullptr;
    d.h_last = bf16 ? s.h_last : nullptr;
    d.store_bf16 = s.store_bf16;
    d.wt_img = (fold && fold->wt_imgs && bwd && !sorted && !nwide && count == count_in) ? fold->wt_imgs[i] : nullptr;
    if (fold && fold->dec_tail == 2 && (sorted || count != count_in)) return MFM_ERR_UNSUPPORTED;
    d.wf_img = (fold && fold->wf_imgs && !bwd && !sorted && !nwide && count == count_in && s.is_dec && (s.h & 3) == 0) ? fold->wf_imgs[i] : nullptr;
    d.wf1_img = d.wf_img ? fold->wf_imgs[count_in + i] : nullptr;        // (second half of the list: W_ih alone)
    MFM_REQUIRE(bf16 || (!s.store_bf16 && !s.h_last), "lstm_seq[%d]: store_bf16 / h_last are taken by the bf16 entry points only", i);
    d.h = s.h; d.Hp = round_up(s.h, 16);
    d.hk4 = round_up(cdiv(s.h, 4), 2);
    d.is_dec = s.is_dec;
    d.block_begin = total;
    total += tiles;
    if (d.Hp / 16 > max_waves) max_waves = d.Hp / 16;
    const size_t HK = (size_t)d.hk4 * 4;
    const size_t need = (bwd ? 2 * 4 * HK * 16 : 2 * HK * 16) * sizeof(float);
    if (need > lds_bytes) lds_bytes = need;
  }
  L.bf16_dot = 1;
  for (int i = 0; i < count; ++i) L.bf16_dot &= (descs[i].bf16_dot != 0);
  if (fold && fold->img_written) *fold->img_written = false;
  if (fold && fold->img_items && !bwd && !bf16 && !sorted && !nwide && fold->n_img_items <= MFM_IMG_MAX && use_small_path(B)) {
    L.n_img = fold->n_img_items;
    for (int i = 0; i < L.n_img; ++i) L.img[i] = fold->img_items[i];
  }
  if (fold && fold->dec_tail == 2) return seq_small_decbwd_head_launch(L, *fold->lat, fold->params, fold->grads, stream);
  if (fold && fold->dec_tail) return seq_small_dectail_launch(L, *fold->lat, fold->params, stream);
  if (fold && !fold->lat && fold->img_items) {      // forward launch with image-writer blocks behind the rows
    const int rc = seq_small_launch(L, false, stream);
    if (rc == MFM_OK && fold->img_written) *fold->img_written = L.n_img > 0;
    return rc;
  }
  if (fold && !fold->lat) {          // images only: the plain one-row launch (one-row tiles: checked by the caller's conditions)
    if ((long)L.count * L.B >= 6L * device_cus()) for (int i = 0; i < L.count; ++i) { L.d[i].wt_img = nullptr; L.d[i].wf_img = nullptr; L.d[i].wf1_img = nullptr; }
    return seq_small_launch(L, bwd, stream);
  }
  if (fold && fold->pr) return seq_small_foldproj_launch(L, *fold->lat, *fold->pr, fold->params, stream);
  if (fold && fold->dr) return seq_small_folddw_launch(L, *fold->lat, *fold->dr, fold->params, fold->grads, stream);
  if (fold) {
    const int rc = seq_small_fold_launch(L, bwd, *fold->lat, fold->params, fold->grads, stream);
    if (rc == MFM_OK && fold->img_written) *fold->img_written = !bwd && L.n_img > 0;
    return rc;
  }
  if (bf16) return seq_bf16_launch(L, bwd, stream);      // bf16 MFMA operands: one kernel family for every batch size
  if (use_small_path(B)) return seq_small_launch(L, bwd, stream);
  // encoders and decoders run different kernels: a mixed call becomes two launches
  for (int kind = 0; kind < 2; ++kind) {
    SeqLaunch K = L;
    K.count = 0;
    int ktotal = 0;
    for (int i = 0; i < L.count; ++i) {
      if ((L.d[i].is_dec != 0) != (kind == 1)) continue;
      K.d[K.count] = L.d[i];
      K.d[K.count].block_begin = ktotal;
      ktotal += tiles;
      ++K.count;
    }
    if (K.count == 0) continue;
    const dim3 grid(ktotal), block(64 * max_waves);
    if (bwd) {
      if (kind) MFM_LAUNCH_TIMED((lstm_seq_kernel<true, 1>), grid, block, lds_bytes, stream, K);
      else MFM_LAUNCH_TIMED((lstm_seq_kernel<true, 0>), grid, block, lds_bytes, stream, K);
    } else {
      if (kind) MFM_LAUNCH_TIMED((lstm_seq_kernel<false, 1>), grid, block, lds_bytes, stream, K);
      else MFM_LAUNCH_TIMED((lstm_seq_kernel<false, 0>), grid, block, lds_bytes, stream, K);
    }
    MFM_LAUNCH_CHECK(bwd ? "lstm_seq_bwd_kernel" : "lstm_seq_fwd_kernel");
  }
  return MFM_OK;
}

// encoder recurrences + their rows' latent chains in one launch (lstm_seq_small.hip); MFM_ERR_UNSUPPORTED: not applicable
int seq_fold_launch(const MfmSeqDesc* descs, int count, int T, int B, bool bwd, const LatentDev& lat, const float* params,
                    float* grads, hipStream_t stream, const float* const* wt_imgs, const WtImgItem* img_items, int n_img_items,
                    bool* img_written) {
  FoldArgs f = {&lat, params, grads, nullptr, nullptr, wt_imgs, img_items, n_img_items, img_written, nullptr, 0};
  return seq_launch(descs, count, T, B, bwd, stream, false, &f);
}
// the forward fold launch with projection role workgroups in front (proj_role_dev.h)
int seq_foldproj_launch(const MfmSeqDesc* descs, int count, int T, int B, const LatentDev& lat, const float* params, ProjRole& pr,
                        hipStream_t stream) {
  FoldArgs f = {&lat, params, nullptr, &pr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0};
  return seq_launch(descs, count, T, B, false, stream, false, &f);
}

}  // namespace mfm

namespace mfm {
// forward launch with a few blocks behind the recurrence rows that write this step's transposed-weight images (lstm_seq_dev.h);
// *written: whether they did (one-row tiles and idle CUs left)
int seq_fwd_img_launch(const MfmSeqDesc* descs, int count, int T, int B, const WtImgItem* items, int n_items, bool* written,
                       hipStream_t stream) {
  FoldArgs f = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, items, n_items, written, nullptr, 0};
  return seq_launch(descs, count, T, B, false, stream, false, &f);
}
// decoder recurrences + the tail blocks of the latent forward chains (lstm_seq_small_dectail_kernel)
int seq_dec_tail_launch(const MfmSeqDesc* descs, int count, int T, int B, const float* const* wf_imgs, const LatentDev& lat,
                        const float* params, hipStream_t stream) {
  FoldArgs f = {&lat, params, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, wf_imgs, 1};
  return seq_launch(descs, count, T, B, false, stream, false, &f);
}
// decoder BPTTs + the head blocks of the latent backward chains (lstm_seq_small_decbwd_head_kernel)
int seq_dec_bwd_head_launch(const MfmSeqDesc* descs, int count, int T, int B, const float* const* wt_imgs, const LatentDev& lat,
                            const float* params, float* grads, hipStream_t stream) {
  FoldArgs f = {&lat, params, grads, nullptr, nullptr, wt_imgs, nullptr, 0, nullptr, nullptr, 2};
  return seq_launch(descs, count, T, B, true, stream, false, &f);
}
// plain BPTT launch whose one-row workgroups take their transposed weights from this step's images (proj_role_dev.h)
int seq_bwd_img_launch(const MfmSeqDesc* descs, int count, int T, int B, const float* const* wt_imgs, hipStream_t stream) {
  FoldArgs f = {nullptr, nullptr, nullptr, nullptr, nullptr, wt_imgs, nullptr, 0, nullptr, nullptr, 0};
  return seq_launch(descs, count, T, B, true, stream, false, &f);
}
// plain forward launch whose one-row decoder workgroups take W_ih + W_hh (steps >= 1) and W_ih (step 0) from this step's
// forward images: wf_imgs[0 .. count) the sums, wf_imgs[count .. 2 count) W_ih
int seq_fwd_wf_launch(const MfmSeqDesc* descs, int count, int T, int B, const float* const* wf_imgs, hipStream_t stream) {
  FoldArgs f = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, wf_imgs, 0};
  return seq_launch(descs, count, T, B, false, stream, false, &f);
}
// the backward fold launch with weight-gradient role workgroups behind the BPTT workgroups (dw_role_dev.h)
int seq_folddw_launch(const MfmSeqDesc* descs, int count, int T, int B, const LatentDev& lat, const float* params, float* grads,
                      DwRole& dr, hipStream_t stream, const float* const* wt_imgs) {
  FoldArgs f = {&lat, params, grads, nullptr, &dr, wt_imgs, nullptr, 0, nullptr, nullptr, 0};
  return seq_launch(descs, count, T, B, true, stream, false, &f);
}
}  // namespace mfm

extern "C" int mfm_lstm_seq_fwd(const MfmSeqDesc* descs, int count, int T, int B, void* stream) {
  return mfm::seq_launch(descs, count, T, B, false, (hipStream_t)stream);
}
extern "C" int mfm_lstm_seq_bwd(const MfmSeqDesc* descs, int count, int T, int B, void* stream) {
  return mfm::seq_launch(descs, count, T, B, true, (hipStream_t)stream);
}
extern "C" int mfm_lstm_seq_fwd_bf16(const MfmSeqDesc* descs, int count, int T, int B, void* stream) {
  return mfm::seq_launch(descs, count, T, B, false, (hipStream_t)stream, true);
}
extern "C" int mfm_lstm_seq_bwd_bf16(const MfmSeqDesc* descs, int count, int T, int B, void* stream) {
  return mfm::seq_launch(descs, count, T, B, true, (hipStream_t)stream, true);
}
