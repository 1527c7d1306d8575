// Shared declarations of the fused plan (include/mfm_hip.h, "The fused MFM_KL_EF training / inference plan"): the plan
// object and what its translation units call in each other.
//   plan_build.hip     parameter index, workspace carving, the latent stack's op / item tables        (mfm::build)
//   plan_forward.hip   the forward chain F0..F4                                                        (mfm::forward)
//   plan_mfn.hip       the Memory Fusion Network's part of both chains (variants MFM_KL / MFM)         (mfm::mfn_forward / mfn_backward)
//   plan_backward.hip  the backward chain B0..B5, the weight-gradient role table                       (mfm::backward)
//   plan.hip           the C ABI: create / options / state / the step entry points / timing
#pragma once
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <new>
#include <vector>

#include "internal.h"
#include "proj_role_dev.h"
#include "dw_role_dev.h"
#include "pack_dev.h"
#include "lstm_seq_dev.h"

namespace mfm {

enum KernelId {
  K_PROJ = 0, K_ENC_FWD, K_LAT_FWD, K_DEC_FWD, K_FC1_FWD, K_MSE, K_FC1_BWD, K_DEC_BWD, K_DEC_DW,
  K_LAT_BWD, K_ENC_BWD, K_ENC_DW, K_ADAM, K_LAT_DW, K_PACK,
  // Memory Fusion Network (variants 1, 2)
  K_MFN_GLUE, K_MFN_ATT_FWD, K_MFN_MEM_FWD, K_MFN_HEADS, K_MFN_MEM_BWD, K_MFN_ATT_BWD, K_MMD, K_COUNT
};

// Index of every tensor group in the reference model's state_dict order (see include/mfm_hip.h):
//   MFM_KL_EF  78 tensors: enc l,a,v | dec l,a,v | ef_encoder | heads | z->f | classifier
//   MFM_KL    104 tensors: enc l,a,v | dec l,a,v | mfn_encoder (32) | heads | z->f | classifier
//   MFM        90 tensors: the same without the logvar heads and the modality mu heads
struct PIdx {
  int enc[4], dec[3];            // encoderLSTM / decoderLSTM blocks: 6 tensors each; enc[3] = ef_encoder (variant 0)
  int mfl[3];                    // MFN LSTMCells: 4 tensors each (weight_ih, weight_hh, bias_ih, bias_hh)
  int att1_1, att1_2, att2_1, att2_2, g1_1, g1_2, g2_1, g2_2;     // MFN Linears (weight; bias = +1)
  int to_z[4], to_lv[4];         // mu / logvar heads in the order l, a, v, y; -1 = absent
  int zf1[4], zf2[4];            // z -> f MLPs, order l, a, v, y
  int y_f1, y_f2;
  int count;
};
PIdx pidx_for(int variant);
enum { W_IH = 0, W_HH = 1, B_IH = 2, B_HH = 3, FC_W = 4, FC_B = 5 };
#define MFM_MAX_NPARAM 104

struct SeqBuf { int64_t gates, hs, cs, wpack; int h, Hp; };

struct TimingPair { hipEvent_t a, b; int kid; };

}  // namespace mfm

struct MfmPlan {
  MfmPlanConfig cfg;
  mfm::PIdx pi;
  int64_t off[MFM_MAX_NPARAM];
  int64_t n_params;
  int D, T, B;
  // sequence encoders: variant 0: l, a, v, early-fusion (n_enc = 4); variants 1, 2: l, a, v + the three MFN LSTMs
  // (n_enc = 6; entries 3..5 have no fc1 head, their cell states feed the attention block); decoders l, a, v
  int n_enc;
  int enc_d[6], enc_xoff[6], enc_h[6], enc_p[6];
  int dec_d[3], dec_h[3], dec_p[3], dec_xoff[3];
  mfm::SeqBuf enc[6], dec[3];
  int64_t dec_dhs[3], dec_init[3], dec_dinit[3], xhat[3], dxhat[3];
  int64_t lat_rec, dh_last[4], yhat, ones, losses;
  // ---- Memory Fusion Network buffers (element offsets into the workspace; variants 1, 2)
  int tot, A2, nzy;                  // sum of MFN hidden sizes, width of cStar, width of the latent's y input
  int64_t dcx[3];                    // d loss / d c_t of the MFN LSTMs [T,B,Hp] (dc_ext of the BPTT)
  int64_t cstar, h1, m1, att, attended, h2, m2, chat, a1, a2, gam1, gam2, mems, mem_out;
  int64_t zero_blk, zero_len;        // cleared by the step's first launch: dcx | zyin | d_hT | dmem | datt
  int64_t dhs_blk, dhs_len;          // the decoders' dH buffers (cleared by the first launch when the fused fc1 kernel runs)
  int64_t zyin, d_hT, dmem, datt;
  int64_t du1, du2, dchat, dh2, dlog, dh1, dcs;
  int64_t lat_seed;                  // variant 2: gradient seed record of the latent backward (d MMD / d z)
  int64_t mmd_scr;                   // variant 2, large B: Gram / kernel matrices of the MMD's GEMM form (-1: row kernel)
  int z_seg[4];                      // variant 2: record offsets of z_l, z_a, z_v, z_y
  const float* gauss;                // variant 2: caller's N(0,1) sample [B, zl+za+zv+zy]
  int64_t ws_floats;
  mfm::LatentDev lat;
  mfm::LatOp lat_ops[MFM_LAT_MAXOPS];
  int64_t lat_ops_off, dbg_off, lat_grd, lat_items_off;
  int lay_f1[4], lay_m1[4], lay_c1, lay_mc;   // record offsets kept for mfm_plan_latent_layout
  std::vector<int> lat_items;       // row-path item tables: forward then backward, [MAXSTAGES][1024][4] each
  // timing
  int timing_mask, timing_every;
  std::vector<mfm::TimingPair> pool;
  size_t pool_used;
  uint64_t calls;
  const float* grads_prezeroed;     // gradient buffer cleared by the forward pass of the running fused step
  int fold_state = 0;               // encoder + latent fold launches (lstm_seq_small.hip): 0 untried, 1 in use, -1 not applicable
  int projfold_state = 0;           // projection role workgroups in the forward fold launch (proj_role_dev.h): 0 / 1 / -1 alike
  int64_t pf_flags = -1;            // their flag words [4][T][16] (u32)
  int64_t wt_img[mfm::MFM_WT_MAX] = {-1, -1, -1, -1, -1, -1, -1, -1, -1};     // transposed-weight images: encoders in plan order, then the 3 decoders (lstm_seq_dev.h)
  unsigned long long wt_call = ~0ull;       // value of `calls` whose forward wrote them
  int64_t wf_img[6] = {-1, -1, -1, -1, -1, -1};     // forward-order images of the decoders' W_ih + W_hh, then of W_ih (lstm_seq_dev.h, wf_img_write)
  unsigned long long wf_call = ~0ull;       // value of `calls` whose encoder launch wrote them
  unsigned long long lat_tail_call = ~0ull; // value of `calls` whose encoder launch left the latent chains' tails to the decoder launch
  int dwfold_state = 0;             // weight-gradient role workgroups in the backward fold launch (dw_role_dev.h): 0 / 1 / -1
  int64_t dw_flags = -1, dw_table = -1;     // stamps [4][T][32] + [4][B]; block table [DWR_TABLE_CAP] int4
  std::vector<int> dw_table_host;   // the table as uploaded (4 ints per block)
  unsigned dw_epoch = 0;            // stamp value of the next backward launch with role workgroups (its own counter: two backward
                                    // calls behind one forward must not see each other's stamps)
  int dw_table_key = -1;            // what it was built for (stage / upstream-gradient form)
  const float* dw_table_ws = nullptr;       // the workspace that holds it
  // ---- bf16 plans (decided once, when the plan is built)
  bool seq_bf16 = false;            // the recurrences run on the bf16 MFMA kernels (lstm_seq_bf16.hip)
  bool st16 = false;                // bf16-RESIDENT saved activations: gates / dA, hs, dH, d x_hat live in HBM as bf16 (round 3)
  int64_t h_last[6];                // st16: fp32 copy of h_{T-1} per encoder [B, Hp] (the latent stack / the MFN heads read it)
  int64_t x16; int x16_ld, x16_off[3];   // st16: bf16 image of the batch [T*B, x16_ld], every modality slice on a 16-column boundary
  int dxh_ld[3];                    // st16: row stride of the bf16 d x_hat buffers
  int64_t fc1_wimg[3];              // st16: scratch for the decoder fc1 weight images (dec_fc1_large.hip)
  bool proj16 = false;              // st16: the projections run on proj_bf16_kernel (which also writes x16)
  mfm::ProjPlan pj;                 // its tile image layout, panel height and pipeline depth
  int64_t pj_wimg, pj_bimg;         // scratch: packed bf16 weight tiles, combined biases
  unsigned long long x16_call = ~0ull;   // value of `calls` for which the forward already produced x16
  int64_t dwb_slabs = -1, dwb_slab_floats = 0;      // st16: scratch for the partial tiles of the one-pass weight-gradient launch (dw_bf16.hip)
  unsigned long long pj_pack_call = ~0ull, fc1_pack_call = ~0ull;   // ... for which the step's pack launch built these images
  unsigned long long fc1_bwd_call = ~0ull;   // value of `calls` for which the forward already produced dH of the decoders (dec_fc1.hip)
  mfm::OptTable* opts = nullptr;    // the MFM_* switches of this plan (common.h): environment at creation + mfm_plan_set_option_str
  // ---- per-plan switches (mfm_plan_set_option, include/mfm_hip.h)
  int opt_handover = 1;             // in-launch hand-overs (role workgroups) allowed
  int64_t opt_timeout_us = 50000;   // how long their consumers spin before they give up
  int64_t opt_guard = -1;           // element offset of the guard word in the gradient buffer (-1: none)
  int opt_fault = 0;                // one-shot fault injection (tests)
  int opt_bf16_dot = 0;             // bf16 plans below the bf16 MFMA kernels' batch size: one-row recurrences on bf16 dot products
  bool ever_handover = false;       // a role-workgroup launch has run on this plan (its status word may be set)
  unsigned* host_status = nullptr;  // 16 words of host-coherent pinned memory (mfm_plan_host_status): [0] / [1] raised by a consumer that gave up
  // device-side state, right behind the plan's loss slots (mfm_plan_state_layout): float offsets relative to `losses`
  static constexpr int ST_STATUS = MFM_LOSS_SLOTS, ST_TICK = MFM_LOSS_SLOTS + 2, ST_DW_TICK = MFM_LOSS_SLOTS + 4;
  unsigned* status_ptr(float* W) const { return reinterpret_cast<unsigned*>(W + losses + ST_STATUS); }
  unsigned* tick_ptr(float* W) const { return reinterpret_cast<unsigned*>(W + losses + ST_TICK); }      // low word of the u64 replay counter
  unsigned* dw_tick_ptr(float* W) const { return reinterpret_cast<unsigned*>(W + losses + ST_DW_TICK); }
  mfm::HoCtl ho_ctl(float* W, float* poison, unsigned bit) const {
    return mfm::HoCtl{status_ptr(W), host_status, poison, opt_timeout_us * 100ll /* 100 MHz wall clock */, bit};
  }
};

namespace mfm {

int build(MfmPlan* P);

// (plan_forward.hip)
__global__ void tick_kernel(unsigned long long* t64, unsigned* t32);
__global__ void guard_propagate_kernel(const unsigned* status, float* guard);
bool stream_capturing(hipStream_t s);
// host part of a hand-over epoch: consecutive launches of one plan -- eager calls and replays of graphs captured at
// different call counts, in any order -- must never carry the same value (epoch = this + replay counter, ho_epoch)
inline unsigned epoch_base(uint64_t calls) { return (unsigned)calls * 0x9E3779B1u; }

struct Timer {
  MfmPlan* P; hipStream_t s; int kid; TimingPair* tp;
  LaunchEvents le; LaunchEvents* prev;
  Timer(MfmPlan* P_, hipStream_t s_, int kid_) : P(P_), s(s_), kid(kid_), tp(nullptr), prev(nullptr) {
    le.launches = -1;
    if (!(P->timing_mask & (1 << kid))) return;
    // sampled: a bracket is two extra packets on the stream (~4.6 us per bracket); timing every step would put that
    // into every step of bench.py's timed region, so only every `timing_every`-th call of the plan is bracketed
    if (P->timing_every > 1 && (P->calls % (uint64_t)P->timing_every) != 0) return;
    if (P->pool_used == P->pool.size()) {
      if (P->pool.size() >= 65536) return;
      TimingPair t; t.kid = -1;
      if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) return;
      P->pool.push_back(t);
    }
    tp = &P->pool[P->pool_used++];
    tp->kid = kid;
    (void)hipEventRecord(tp->a, s);
    // a single launch inside this region takes both events as its own dispatch timestamps (common.h, MFM_LAUNCH_TIMED)
    le.a = tp->a; le.b = tp->b; le.launches = 0;
    prev = tls_launch_events;
    tls_launch_events = &le;
  }
  ~Timer() {
    if (!tp) return;
    tls_launch_events = prev;
    // no launch went through the macro: the plain bracket; several did: from the first kernel's own begin to behind the last
    if (le.launches != 1) (void)hipEventRecord(tp->b, s);
  }
};

#define RUN(kid, expr)                        \
  do {                                        \
    Timer _t(P, s, kid);                      \
    int _rc = (expr);                         \
    if (_rc != MFM_OK) return _rc;            \
  } while (0)

MfmSeqDesc seq_desc(const MfmPlan* P, const SeqBuf& sb, int pbase, const float* params, float* W, bool dec);
inline const float* PW(const MfmPlan* P, const float* params, int idx) { return params + P->off[idx]; }

struct LossW { float disc; int gen_on; float reg; int write_disc; };

struct ExtGrads {           // upstream gradients supplied by the caller (autograd module path)
  const float* d_xhat[3];
  const float* d_yhat;
  const float* d_reg;       // device scalar
};

bool mfn_heads_desc(const MfmPlan* P, const float* params, float* W, MfnHeadsDev& H);
int mfn_forward(MfmPlan* P, const float* params, int train, uint64_t seed, float* W, hipStream_t s);
int mfn_backward(MfmPlan* P, const float* params, float* W, float* grads, hipStream_t s, std::vector<MfmGemmDesc>& tail);
int forward(MfmPlan* P, const float* params, const float* x, const void* y, int train, uint64_t seed, float* W, float* xhat_out[3],
            float* yhat_out, float* losses_out, hipStream_t s, float* grads_to_zero = nullptr);
void dA_gemms(const MfmPlan* P, const SeqBuf& sb, int pb, float* W, float* grads, std::vector<MfmGemmDesc>& out, const float* xin,
              int64_t ldx, int kin, bool dec, bool only_init = false);
int dw_role_build(MfmPlan* P, const std::vector<MfmGemmDesc>& all, float* W, int key, hipStream_t s, DwRole* out);
int backward(MfmPlan* P, const float* params, const float* x, const void* y, int stage, float* W, float* grads, hipStream_t s,
             const ExtGrads* ext = nullptr, const LossW* lw = nullptr);

}  // namespace mfm
