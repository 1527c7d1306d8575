// The fused MFM_KL_EF step: one host call enqueues the whole forward / backward / Adam chain
// (11 launches, no memsets) on one HIP stream, with no host work or synchronisation between
// kernels.  Replaces MFM_KL_EF.forward (reference mfm_model.py:619-660), the joint loss and
// loss.backward()/optimizer.step() of train_mfm.train (mfm_mosi.py:424-442).
//
// Launch chain (F = forward, B = backward):
//   F0 grouped GEMM   x_t W_ih^T + b_ih + b_hh for all t, 4 encoders        -> gates_e
//                     (the same launch clears the loss slots and, in the fused step, the gradient buffer)
//   F1 lstm_seq fwd   4 encoder recurrences (persistent, weights in VGPRs)   -> gates/hs/cs
//   F2 latent fwd     enc.fc1, mu/logvar heads, z->f MLPs, classifier, KLD, L1|CE
//   F3 lstm_seq fwd   3 decoder recurrences
//   F4 grouped GEMM   decoder fc1 -> x_hat, with the squared-error epilogue  -> 3 reconstruction losses, d x_hat
//   B0 grouped GEMM   dH = dx_hat Wfc                                        (3 problems)
//   B1 lstm_seq bwd   3 decoder BPTTs                                        -> dA, d h_init
//   B3 latent bwd
//   B4 lstm_seq bwd   4 encoder BPTTs
//   B5 grouped GEMM   EVERY weight gradient: dWfc/dbfc, the 22 latent dW, encoder and decoder
//                     dW_ih/dW_hh/db over dA                                 (49 problems, one launch)
//   A  adam           fused, one flat buffer
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
static PIdx pidx_for(int variant) {
  PIdx p;
  memset(&p, 0xff, sizeof(p));      // -1 everywhere
  p.enc[0] = 0; p.enc[1] = 6; p.enc[2] = 12;
  p.dec[0] = 18; p.dec[1] = 24; p.dec[2] = 30;
  if (variant == 0) {
    p.enc[3] = 36;
    p.to_z[3] = 42; p.to_lv[3] = 44; p.to_z[0] = 46; p.to_z[1] = 48; p.to_z[2] = 50;
    p.to_lv[0] = 52; p.to_lv[1] = 54; p.to_lv[2] = 56;
    p.zf1[3] = 58; p.zf2[3] = 60; p.zf1[0] = 62; p.zf2[0] = 64; p.zf1[1] = 66; p.zf2[1] = 68; p.zf1[2] = 70; p.zf2[2] = 72;
    p.y_f1 = 74; p.y_f2 = 76; p.count = 78;
    return p;
  }
  p.mfl[0] = 36; p.mfl[1] = 40; p.mfl[2] = 44;
  p.att1_1 = 48; p.att1_2 = 50; p.att2_1 = 52; p.att2_2 = 54; p.g1_1 = 56; p.g1_2 = 58; p.g2_1 = 60; p.g2_2 = 62;
  // 64..67: mfn_encoder.out_fc1 / out_fc2 -- in the state_dict, unused by forward (reference mfm_model.py:133-137,199)
  p.to_z[3] = 68;
  int next = 70;
  if (variant == 1) {
    p.to_lv[3] = 70; p.to_z[0] = 72; p.to_z[1] = 74; p.to_z[2] = 76; p.to_lv[0] = 78; p.to_lv[1] = 80; p.to_lv[2] = 82;
    next = 84;
  }
  p.zf1[3] = next; p.zf2[3] = next + 2;
  for (int e = 0; e < 3; ++e) { p.zf1[e] = next + 4 + 4 * e; p.zf2[e] = next + 6 + 4 * e; }
  p.y_f1 = next + 16; p.y_f2 = next + 18; p.count = next + 20;
  return p;
}
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

static int64_t carve(int64_t& cursor, int64_t n) {
  const int64_t at = cursor;
  cursor = round_up64(cursor + n, 64);   // 256-byte granules
  return at;
}

static void add_op(LatOp* ops, LatentDev& L, int stage, int chain, int in_off, int out_off, int K, int N, int64_t w_off,
                   int64_t b_off, int relu, int mask_off, float p) {
  LatOp& o = ops[L.nops++];
  o.in_off = in_off; o.out_off = out_off; o.K = K; o.N = N; o.w_off = w_off; o.b_off = b_off;
  o.relu = relu; o.mask_off = mask_off; o.drop_p = p; o.stage = stage; o.chain = chain;
}

static int build(MfmPlan* P) {
  const MfmPlanConfig& c = P->cfg;
  const PIdx& pi = P->pi;
  const int V = c.variant;
  P->T = c.T; P->B = c.B;
  P->D = c.d_l + c.d_a + c.d_v;
  const int ze = c.zl + c.za + c.zv;
  const int dd[3] = {c.d_l, c.d_a, c.d_v};
  const int dx[3] = {0, c.d_l, c.d_l + c.d_a};
  const int fm[3] = {c.fl, c.fa, c.fv};
  const int mh[3] = {c.hl, c.ha, c.hv};
  int64_t cur = 0;
  const int64_t TB = (int64_t)c.T * c.B;
  P->n_enc = (V == 0) ? 4 : 6;
  // bf16 plans: kernel family of the recurrences, and whether the saved activations are bf16-resident.  bf16-resident needs the
  // bf16 recurrences for every LSTM (h <= 128: the step-by-step path of wider ones is fp32) and shapes the one-pass
  // weight-gradient kernel takes (dw_bf16.hip); MFM_BF16_STORE=0 keeps the round-2 form (fp32 buffers, rounding on load).
  P->seq_bf16 = c.precision && bf16_seq_pays(c.B);
  {
    // default from T*B = 3840 rows (B = 192 at T = 20: where the bf16 recurrences start).  Measured at the MOSI sizes
    // (bf16-resident vs fp32-stored, ms per step; B = 192 / 256 / 384 / 512 / 768 / 1024): 0.366 vs 0.377, 0.377 vs 0.418,
    // 0.406 vs 0.471, 0.418 vs 0.521, 0.450 vs 0.603, 0.481 vs 0.706 (round 2: crossover at T*B = 16384; then proj_bf16.hip, the
    // 64-row decoder fc1 and whole rounds of workgroups in the one-pass weight-gradient launch); MFM_BF16_STORE=1 forces it
    // on for every size, =0 off
    const char* se = opt_get("MFM_BF16_STORE");
    long st_minrows = 3840;
    if (const char* e = opt_get("MFM_BF16_STORE_MINROWS")) st_minrows = atol(e);
    bool ok = P->seq_bf16 && !opt_get("MFM_SEQ_STEPWISE") && (se ? atoi(se) != 0 : TB >= st_minrows);
    const int Dp = round_up(c.d_l, 16) + round_up(c.d_a, 16) + round_up(c.d_v, 16);
    int hmax = 0, np_max = 0;
    for (int e = 0; e < P->n_enc; ++e) {
      int h, xc;
      if (e < 3) { h = (e == 0 ? c.zl : (e == 1 ? c.za : c.zv)); xc = round_up(dd[e], 16); }
      else if (V == 0) { h = ze; xc = Dp; }
      else { h = mh[e - 3]; xc = round_up(dd[e - 3], 16); }
      hmax = std::max(hmax, h);
      np_max = std::max(np_max, xc + round_up(h, 16));
    }
    for (int m = 0; m < 3; ++m) hmax = std::max(hmax, c.fy + fm[m]);
    ok = ok && hmax <= MFM_SEQ_MAX_RESIDENT_H && np_max <= 576 && (32 * (96 + np_max) / 8 + 511) / 512 <= 6;
    ok = ok && TB * 4 * round_up(hmax, 16) * 2 < ((int64_t)1 << 31) && TB * Dp * 2 < ((int64_t)1 << 31);
    P->st16 = ok;
    if (opt_get("MFM_PLAN_DEBUG")) fprintf(stderr, "[mfm plan] bf16: recurrences on the bf16 kernels %d, bf16-resident activations %d\n", (int)P->seq_bf16, (int)P->st16);
  }
  const int ESH = P->st16 ? 2 : 1;          // bf16-resident buffers take half the floats
  for (int e = 0; e < P->n_enc; ++e) {
    if (e < 3) { P->enc_d[e] = dd[e]; P->enc_xoff[e] = dx[e]; P->enc_h[e] = (e == 0 ? c.zl : (e == 1 ? c.za : c.zv)); P->enc_p[e] = pi.enc[e]; }
    else if (V == 0) { P->enc_d[e] = P->D; P->enc_xoff[e] = 0; P->enc_h[e] = ze; P->enc_p[e] = pi.enc[3]; }
    else { P->enc_d[e] = dd[e - 3]; P->enc_xoff[e] = dx[e - 3]; P->enc_h[e] = mh[e - 3]; P->enc_p[e] = pi.mfl[e - 3]; }
    SeqBuf& s = P->enc[e];
    s.h = P->enc_h[e]; s.Hp = round_up(s.h, 16);
    s.gates = carve(cur, TB * 4 * s.Hp / ESH);
    s.hs = carve(cur, TB * s.Hp / ESH);
    s.cs = carve(cur, TB * s.Hp);
    s.wpack = c.precision ? carve(cur, mfm_lstm_pack_bytes(s.h, 0) / 4) : -1;
    P->h_last[e] = P->st16 ? carve(cur, (int64_t)c.B * s.Hp) : -1;
    if (e < 4) P->dh_last[e] = -1;
    if (e < 3 || V == 0) P->dh_last[e] = carve(cur, (int64_t)c.B * P->enc_h[e]);
  }
  for (int m = 0; m < 3; ++m) {
    P->dec_d[m] = dd[m]; P->dec_h[m] = c.fy + fm[m]; P->dec_p[m] = pi.dec[m]; P->dec_xoff[m] = dx[m];
    SeqBuf& s = P->dec[m];
    s.h = P->dec_h[m]; s.Hp = round_up(s.h, 16);
    s.gates = carve(cur, TB * 4 * s.Hp / ESH);
    s.hs = carve(cur, TB * s.Hp / ESH);
    s.cs = carve(cur, TB * s.Hp);
    s.wpack = c.precision ? carve(cur, mfm_lstm_pack_bytes(s.h, 1) / 4) : -1;
    P->dec_init[m] = carve(cur, (int64_t)c.B * s.h);
    P->dec_dinit[m] = carve(cur, (int64_t)c.B * s.h);
    P->xhat[m] = carve(cur, TB * dd[m]);
    // st16: d x_hat as bf16 with rows padded to 8 columns (16-byte rows; the pad columns are never written and stay zero)
    P->dxh_ld[m] = P->st16 ? round_up(dd[m], 8) : dd[m];
    P->dxhat[m] = carve(cur, TB * P->dxh_ld[m] / ESH);
  }
  P->dhs_blk = cur;                                // one block: the fused fc1 kernel adds into it (dec_fc1.hip), zero span 3
  for (int m = 0; m < 3; ++m) P->dec_dhs[m] = carve(cur, TB * P->dec[m].Hp / ESH);
  P->dhs_len = cur - P->dhs_blk;
  P->x16 = -1; P->x16_ld = 0;
  if (P->st16) {
    int at = 0;
    for (int m = 0; m < 3; ++m) { P->x16_off[m] = at; at += round_up(dd[m], 16); }
    P->x16_ld = at;
    P->x16 = carve(cur, TB * P->x16_ld / 2);
    for (int m = 0; m < 3; ++m) P->fc1_wimg[m] = carve(cur, (int64_t)(dec_fc1_large_wimg_bytes(dd[m]) + 3) / 4);
    // partial tiles of the one-pass weight-gradient launch (slab form, dw_bf16.hip): 69 MB per round of workgroups
    P->dwb_slab_floats = dw_bf16_scratch_floats(TB);
    P->dwb_slabs = carve(cur, P->dwb_slab_floats);
    // the projections of this plan: proj_bf16.hip when its panel fits the LDS (MFM_PROJ16=0: gemm_panel / tiled GEMM)
    PanelLaunch PL;
    memset(&PL, 0, sizeof(PL));
    PL.M = (int)TB; PL.K = P->D; PL.ngroups = P->n_enc;
    for (int e = 0; e < P->n_enc && e < MFM_PANEL_MAXG; ++e) {
      PanelGroup& G = PL.g[e];
      G.n = 4 * P->enc[e].Hp; G.seg = P->enc[e].Hp; G.seg_valid = P->enc[e].h; G.k_off = P->enc_xoff[e]; G.k_len = P->enc_d[e];
    }
    const char* pe = opt_get("MFM_PROJ16");
    P->proj16 = (!pe || atoi(pe) != 0) && P->n_enc <= MFM_PANEL_MAXG && TB * P->D < ((int64_t)1 << 29) && proj_bf16_plan(PL, &P->pj);
    if (P->proj16) {
      P->pj_wimg = carve(cur, (int64_t)P->pj.ntiles * 2048);
      P->pj_bimg = carve(cur, P->pj.nbias);
    }
    if (opt_get("MFM_PLAN_DEBUG")) fprintf(stderr, "[mfm plan] bf16-resident projections: proj_bf16_kernel %d (%d tiles, %d-row panels, %d stages)\n", (int)P->proj16, P->pj.ntiles, P->pj.BM, P->pj.S);
  }
  // ---- Memory Fusion Network (variants 1, 2): every [T*B, .] tensor of the attention block and the memory recurrence
  P->tot = P->A2 = P->nzy = 0;
  if (V != 0) {
    P->tot = c.hl + c.ha + c.hv; P->A2 = 2 * P->tot;
    P->nzy = (V == 1) ? 2 * c.zy : c.zy;          // [mu_y | logvar_y] or z_y: the latent stack's fourth input
    const int M = c.mem_dim;
    P->cstar = carve(cur, TB * P->A2); P->att = carve(cur, TB * P->A2); P->attended = carve(cur, TB * P->A2);
    P->h1 = carve(cur, TB * c.nn1); P->m1 = carve(cur, TB * c.nn1);
    P->h2 = carve(cur, TB * c.nn2); P->m2 = carve(cur, TB * c.nn2);
    P->chat = carve(cur, TB * M);
    P->a1 = carve(cur, TB * c.g1); P->a2 = carve(cur, TB * c.g2);
    P->gam1 = carve(cur, TB * M); P->gam2 = carve(cur, TB * M); P->mems = carve(cur, TB * M);
    P->mem_out = carve(cur, (int64_t)c.B * M);
    P->zero_blk = cur;
    for (int m = 0; m < 3; ++m) P->dcx[m] = carve(cur, TB * P->enc[3 + m].Hp);   // the fused attention backward adds into these
    P->zyin = carve(cur, (int64_t)c.B * P->nzy);
    P->d_hT = carve(cur, (int64_t)c.B * P->tot);
    P->dmem = carve(cur, (int64_t)c.B * M);
    P->datt = carve(cur, TB * P->A2);
    P->zero_len = cur - P->zero_blk;               // carve() keeps 64-float granules: a multiple of 4
    P->du1 = carve(cur, TB * c.g1); P->du2 = carve(cur, TB * c.g2); P->dchat = carve(cur, TB * M);
    P->dh2 = carve(cur, TB * c.nn2); P->dlog = carve(cur, TB * P->A2); P->dh1 = carve(cur, TB * c.nn1);
    P->dcs = carve(cur, TB * P->A2);
  }
  // ---- latent record layout (every segment starts on a multiple of 4 floats)
  LatentDev& L = P->lat;
  memset(&L, 0, sizeof(L));
  int rs = 0;
  auto seg = [&](int n) { const int at = rs; rs += round_up(n, 4); return at; };
  const int zn[4] = {c.zl, c.za, c.zv, c.zy};
  const int fn[4] = {c.fl, c.fa, c.fv, c.fy};
  // inputs of the stack: last hidden state of the modality encoders, and for y the early-fusion encoder's
  // (variant 0) or the precomputed heads on the MFN output (variants 1, 2: [mu_y | logvar_y] / z_y)
  const int in_n[4] = {c.zl, c.za, c.zv, V == 0 ? ze : P->nzy};
  int last_off[4], f1_off[4], m1_off[4];
  const int nfc = (V == 0) ? 4 : 3;                 // encoder fc1 heads inside the stack
  int c1_off = 0, mc_off = 0;
  // chain by chain (l, a, v, y): every segment a modality's layers read or write is contiguous, so that a workgroup that
  // runs one chain of a row (LatentDev::nch) saves / restores one range of the record
  for (int e = 0; e < 4; ++e) {
    L.ch_lo[e] = rs;
    L.in_off[e] = seg(in_n[e]); L.enc_n[e] = in_n[e];
    last_off[e] = (e < nfc) ? seg(in_n[e]) : -1;
    L.z_n[e] = zn[e];
    if (V == 2) L.mu_off[e] = (e < 3) ? last_off[e] : L.in_off[3];          // z = the encoder output itself
    else if (V == 1 && e == 3) L.mu_off[e] = L.in_off[3];
    else L.mu_off[e] = seg(zn[e]);
    if (V == 2) L.lv_off[e] = 0;
    else if (V == 1 && e == 3) L.lv_off[e] = L.in_off[3] + c.zy;
    else L.lv_off[e] = seg(zn[e]);
    f1_off[e] = seg(fn[e]); m1_off[e] = seg(fn[e]);
    L.f_off[e] = seg(fn[e]); L.f_n[e] = fn[e];
    if (e == 3) {
      c1_off = seg(c.fy); mc_off = seg(c.fy);
      L.yhat_off = seg(c.output_dim); L.od = c.output_dim;
    }
    L.ch_hi[e] = rs;
  }
  L.rec_size = rs;
  for (int e = 0; e < 4; ++e) { P->lay_f1[e] = f1_off[e]; P->lay_m1[e] = m1_off[e]; P->z_seg[e] = L.mu_off[e]; }
  P->lay_c1 = c1_off; P->lay_mc = mc_off;
  const int64_t* o = P->off;
  int st = 0;
  // encoder fc1 (mfm_model.py:60-61).  Batches beyond the row kernels' range (staged kernels, latent.hip) give the
  // early-fusion encoder's fc1 a stage of its own: the four heads together are the largest weight span (89 KB at the MOSI
  // sizes), alone it is 58 KB, and the LDS that frees doubles the rows a workgroup carries (backward 4 -> 8).
  const int lat_row_maxb = opt_get("MFM_LATENT_ROW_MAXB") ? atoi(opt_get("MFM_LATENT_ROW_MAXB")) : 256;   // tuning override
  bool split0 = V == 0 && c.B > lat_row_maxb && c.B > 4 * device_cus();   // (up to 4 rows x CUs one round of 4-row workgroups does)
  if (const char* e = opt_get("MFM_LATENT_SPLIT0")) split0 = V == 0 && atoi(e) != 0;
  for (int e = 0; e < nfc; ++e) {
    if (split0 && e == 3) ++st;
    add_op(P->lat_ops, L, st, e, L.in_off[e], last_off[e], in_n[e], in_n[e], o[pi.enc[e] + FC_W], o[pi.enc[e] + FC_B], 0, -1, 0.f);
  }
  ++st;
  // mu heads (mfm_model.py:630-639 / 737-744).  The logvar heads only feed the KLD, nothing downstream waits
  // for them, so they ride along with the classifier's first layer (the row kernels give every
  // thread one work item per stage: 4*(16+152) output quads and 4*(16+240)/4 input groups still fit 1024).
  if (V != 2) {
    for (int e = 0; e < nfc; ++e)
      add_op(P->lat_ops, L, st, e, last_off[e], L.mu_off[e], in_n[e], zn[e], o[pi.to_z[e]], o[pi.to_z[e] + 1], 0, -1, 0.f);
    ++st;
  }
  // z -> f MLPs (mfm_model.py:644-647)
  const float pd[4] = {c.drop_zl, c.drop_za, c.drop_zv, c.drop_zy};
  for (int e = 0; e < 4; ++e)
    add_op(P->lat_ops, L, st, e, L.mu_off[e], f1_off[e], zn[e], fn[e], o[pi.zf1[e]], o[pi.zf1[e] + 1], 1, m1_off[e], pd[e]);
  ++st;
  for (int e = 0; e < 4; ++e)
    add_op(P->lat_ops, L, st, e, f1_off[e], L.f_off[e], fn[e], fn[e], o[pi.zf2[e]], o[pi.zf2[e] + 1], 1, -1, 0.f);
  ++st;
  // classifier (mfm_model.py:657); its first stage also carries the logvar heads
  add_op(P->lat_ops, L, st, 3, L.f_off[3], c1_off, c.fy, c.fy, o[pi.y_f1], o[pi.y_f1 + 1], 1, mc_off, c.drop_y);
  if (V != 2)
    for (int e = 0; e < nfc; ++e)
      add_op(P->lat_ops, L, st, e, last_off[e], L.lv_off[e], in_n[e], zn[e], o[pi.to_lv[e]], o[pi.to_lv[e] + 1], 0, -1, 0.f);
  ++st;
  add_op(P->lat_ops, L, st, 3, c1_off, L.yhat_off, c.fy, c.output_dim, o[pi.y_f2], o[pi.y_f2 + 1], 0, -1, 0.f);
  ++st;
  L.nstages = st;
  {
    int s = 0;
    L.stage_begin[0] = 0;
    for (int i = 0; i < L.nops; ++i)
      while (P->lat_ops[i].stage > s) L.stage_begin[++s] = i;
    L.stage_begin[L.nstages] = L.nops;
    for (int st = 0; st < L.nstages; ++st) {
      int an = 0, ak = 0;
      for (int i = L.stage_begin[st]; i < L.stage_begin[st + 1]; ++i) {
        P->lat_ops[i].pfx_n = an; P->lat_ops[i].pfx_k = ak;
        an += P->lat_ops[i].N; ak += P->lat_ops[i].K;
      }
    }
  }
  L.has_logvar = (V != 2) ? 1 : 0;
  L.B = c.B;
  L.loss_kind = c.loss_kind;
  // LDS weight panel: the tensors of one stage are expected to be contiguous in the flat buffer
  // (engine.py FlatLayout groups them); the span [min offset, max end) is copied linearly.
  int panel = 0;
  for (int st = 0; st < L.nstages; ++st) {
    int64_t lo = INT64_MAX, hi = 0;
    for (int i = L.stage_begin[st]; i < L.stage_begin[st + 1]; ++i) {
      const LatOp& op = P->lat_ops[i];
      lo = std::min(lo, std::min(op.w_off, op.b_off));
      hi = std::max(hi, std::max(op.w_off + (int64_t)op.N * op.K, op.b_off + (int64_t)op.N));
    }
    lo = lo / 4 * 4;
    int64_t len = round_up64(hi - lo, 4);
    if (lo + len > P->n_params) len = (P->n_params - lo) / 4 * 4;
    L.span_off[st] = lo;
    L.span_len[st] = (len > INT32_MAX) ? INT32_MAX : (int)len;
    if (L.span_len[st] > panel) panel = L.span_len[st];
  }
  // rows per workgroup: small batches want many workgroups, large ones fewer atomics
  const size_t LDS_BUDGET = 150 * 1024;
  int R = (c.B <= 64) ? 4 : ((c.B <= 1024) ? 8 : 16);
  if (const char* e = opt_get("MFM_LATENT_ROWS")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) R = v; }   // tuning override
  if (((size_t)panel + 2 * (size_t)rs) * sizeof(float) <= LDS_BUDGET) {
    L.wpanel = panel;
    while (R > 1 && (2 * (size_t)R * rs + panel) * sizeof(float) > LDS_BUDGET) R >>= 1;
  } else {
    L.wpanel = 0;   // stage tensors not contiguous / too large to stage: kernels read them from L2
    while (R > 1 && 2 * (size_t)R * rs * sizeof(float) > LDS_BUDGET) R >>= 1;
  }
  L.rows_per_wg = R;
  // the staged kernels' products on the fp32 MFMA (latent.hip) when the weight panel is staged and every layer's K is a
  // multiple of 4; MFM_LATENT_MFMA=0 keeps the quad form
  {
    bool ok = L.wpanel > 0 && R <= 16;
    for (int i = 0; i < L.nops && ok; ++i) ok = (P->lat_ops[i].K & 3) == 0 && ((P->lat_ops[i].w_off - L.span_off[P->lat_ops[i].stage]) & 3) == 0;
    if (const char* e = opt_get("MFM_LATENT_MFMA")) ok = ok && atoi(e) != 0;
    L.mfma = ok ? 1 : 0;
  }
  // the forward keeps ONE record per row in LDS (the backward two), so it can take more rows per workgroup: a workgroup's
  // time is mostly the six stage spans it streams from L2 (140 KB, ~17 of ~30 us at 4 rows), not the rows' arithmetic
  {
    // (measured, profiles/r02_latent_rows.txt: a round of 4-row workgroups 31 us, of 8-row ones 43 us), so the rows double
    // while the launch would otherwise need more than one round of workgroups
    int Rf = R;
    const int want = opt_get("MFM_LATENT_ROWS_FWD") ? atoi(opt_get("MFM_LATENT_ROWS_FWD")) : 0;   // tuning override
    while (Rf < 16 && (want ? Rf < want : cdiv(c.B, Rf) > device_cus()) && ((size_t)2 * Rf * rs + L.wpanel) * sizeof(float) <= LDS_BUDGET)
      Rf <<= 1;
    L.rows_fwd = Rf;
    if (opt_get("MFM_PLAN_DEBUG")) fprintf(stderr, "[mfm plan] latent: rec_size %d floats, weight panel %d floats, rows per workgroup bwd %d fwd %d\n", rs, L.wpanel, R, Rf);
  }
  // Latency path (latent.hip, row kernels): one row per workgroup while that still fits the chip in one
  // wave of workgroups and every layer meets the vector-load shape requirements.
  {
    bool ok = c.B <= lat_row_maxb && !split0 && (size_t)2 * rs * sizeof(float) <= 24 * 1024 && P->n_params < (1ll << 31);
    ok = ok && (in_n[0] + in_n[1] + in_n[2] + in_n[3] <= MFM_LAT_ROW_THREADS);     // prologue: one input element per thread
    for (int i = 0; i < L.nops && ok; ++i) {
      const LatOp& op = P->lat_ops[i];
      ok = (op.K % 4 == 0) && op.K >= 4 && op.K <= 128 && op.N <= 128 && (op.w_off % 4 == 0);
    }
    for (int st = 0; st < L.nstages && ok; ++st) {
      int sn = 0, sk = 0;
      for (int i = L.stage_begin[st]; i < L.stage_begin[st + 1]; ++i) { sn += P->lat_ops[i].N; sk += P->lat_ops[i].K; }
      ok = 4 * sn <= 1024 && 4 * sk <= 1024;      // one work item per thread and stage
    }
    if (const char* e = opt_get("MFM_LATENT_PATH")) { if (!strcmp(e, "staged")) ok = false; }
    L.row_path = ok ? 1 : 0;
  }
  // row path: the work item of thread t in stage s is static, so it is tabulated here once (encoding: latent.hip).
  // Chains: at small batches (B * 4 <= CUs) every row's four modality chains get a workgroup each (the forward launch is
  // bound by what ONE CU can stream from L2, ~14 B/clk: 228 KB of weights per row-workgroup = 6.8 us of its 14 us); the
  // tables then exist per chain [chain][stage][thread], chain c seeing only its own layers.  MFM_LATENT_CHAINS=0 disables.
  const int NT = MFM_LAT_ROW_THREADS;
  const size_t TABN = (size_t)4 * MFM_LAT_MAXSTAGES * NT * 4;        // ints per direction
  P->lat_items.assign(2 * TABN, 0);
  L.nch = 1;
  if (L.row_path) {
    bool chains = 4 * c.B <= device_cus();
    if (const char* e = opt_get("MFM_LATENT_CHAINS")) chains = chains && atoi(e) != 0;
    L.nch = chains ? 4 : 1;
    int* fw = P->lat_items.data();
    int* bw = fw + TABN;
    for (int ch = 0; ch < L.nch; ++ch)
      for (int st = 0; st < L.nstages; ++st) {
        const int ob = L.stage_begin[st], oe = L.stage_begin[st + 1];
        // the layers of this stage this workgroup kind runs, with their own prefix sums
        std::vector<int> sel, pn, pk;
        int sn = 0, sk = 0;
        for (int i = ob; i < oe; ++i) {
          if (L.nch > 1 && P->lat_ops[i].chain != ch) continue;
          sel.push_back(i); pn.push_back(sn); pk.push_back(sk);
          sn += P->lat_ops[i].N; sk += P->lat_ops[i].K;
        }
        L.nitems_fwd_c[ch][st] = 4 * sn;
        L.nitems_bwd_c[ch][st] = 4 * sk;
        if (L.nch == 1) { L.nitems_fwd[st] = 4 * sn; L.nitems_bwd[st] = 4 * sk; }
        for (int t = 0; t < NT; ++t) {
          int* ef = fw + (((size_t)ch * MFM_LAT_MAXSTAGES + st) * NT + t) * 4;
          int* eb = bw + (((size_t)ch * MFM_LAT_MAXSTAGES + st) * NT + t) * 4;
          if (sel.empty()) { ef[0] = ef[1] = ef[2] = ef[3] = 0; eb[0] = eb[1] = eb[2] = eb[3] = 0; ef[2] = 4 << 16; eb[1] = 4 | (1 << 8); continue; }
          {   // forward: quad (n, q) -> output column n of op o
            const bool live = t < 4 * sn;
            const int item = std::min(t, 4 * sn - 1) >> 2;
            size_t si = 0;
            while (si + 1 < sel.size() && item >= pn[si + 1]) ++si;
            const int o = sel[si];
            const LatOp& op = P->lat_ops[o];
            const int n = item - pn[si];
            ef[0] = (int)(op.w_off + (int64_t)n * op.K);
            ef[1] = (int)(op.b_off + n);
            ef[2] = op.in_off | (op.K << 16);
            ef[3] = (op.out_off + n) | (o << 16) | ((op.relu ? 1 : 0) << 24) | ((op.mask_off >= 0 ? 1 : 0) << 25) |
                    ((live ? 1 : 0) << 26);
          }
          {   // backward: 16 lanes (kc, l) -> input columns kc..kc+3 of op o
            const bool live = t < 4 * sk;
            const int col = (std::min(t, 4 * sk - 1) >> 4) * 4;
            size_t si = 0;
            while (si + 1 < sel.size() && col >= pk[si + 1]) ++si;
            const int o = sel[si];
            const LatOp& op = P->lat_ops[o];
            const int kc = col - pk[si];
            eb[0] = (int)(op.w_off + kc);
            eb[1] = op.K | (op.N << 8);
            eb[2] = op.out_off | ((op.in_off + kc) << 16);
            // the layer that PRODUCED these input columns: its relu / dropout mask is applied to the gradient
            // as it is accumulated (they are linear, so masking each contribution == masking the sum)
            int prelu = 0, pmask = 0;
            for (int pi = 0; pi < ob; ++pi) {
              const LatOp& pr = P->lat_ops[pi];
              const int idx = op.in_off + kc;
              if (idx >= pr.out_off && idx < pr.out_off + pr.N) {
                prelu = pr.relu ? 1 : 0;
                pmask = pr.mask_off >= 0 ? pr.mask_off + (idx - pr.out_off) + 1 : 0;
              }
            }
            eb[3] = (live ? 1 : 0) | (prelu << 1) | (pmask << 2);
          }
        }
      }
    // MFM_LATENT_PRE=1 (opt-in): chain workgroups of 512 threads that request the weights four stages ahead instead of one
    // -- measured no faster (13.5 vs 13.8 us forward: the stages are not waiting for weights), profiles/r02_latent_chains.txt
    L.pre = 0;
    if (const char* e = opt_get("MFM_LATENT_PRE")) L.pre = (atoi(e) != 0 && L.nch > 1 && L.nstages <= 6) ? 1 : 0;
    for (int ch = 0; ch < L.nch && L.pre; ++ch)
      for (int st = 0; st < L.nstages; ++st)
        if (L.nitems_fwd_c[ch][st] > 512 || L.nitems_bwd_c[ch][st] > 512) L.pre = 0;
    // chain workgroups whose widest stage fits 512 threads are launched with 512: half the item table to copy in the prologue
    // (16 bytes per thread and stage), half the waves to walk through every barrier
    L.row_threads = MFM_LAT_ROW_THREADS;
    if (L.nch > 1) {
      int mx = 0;
      for (int ch = 0; ch < L.nch; ++ch)
        for (int st = 0; st < L.nstages; ++st) mx = std::max(mx, std::max(L.nitems_fwd_c[ch][st], L.nitems_bwd_c[ch][st]));
      int in_sum = 0;
      for (int e = 0; e < 4; ++e) in_sum += L.enc_n[e];
      if (mx <= 512 && in_sum <= 512 && !(opt_get("MFM_LATENT_512") && atoi(opt_get("MFM_LATENT_512")) == 0)) L.row_threads = 512;
    }
    if (L.nch > 1)       // whole-stage counts (bias-gradient loops of the backward walk all layers of a stage)
      for (int st = 0; st < L.nstages; ++st) {
        int sn = 0, sk = 0;
        for (int i = L.stage_begin[st]; i < L.stage_begin[st + 1]; ++i) { sn += P->lat_ops[i].N; sk += P->lat_ops[i].K; }
        L.nitems_fwd[st] = 4 * sn; L.nitems_bwd[st] = 4 * sk;
      }
  }

  P->lat_ops_off = carve(cur, (int64_t)(sizeof(P->lat_ops) / (sizeof(float))));
  P->dbg_off = carve(cur, 128);     // 64 x u64 debug timestamps
  P->pf_flags = (V == 0) ? carve(cur, (int64_t)4 * P->T * PROJ_ROLE_FLAGS) : -1;
  if ((long)(P->n_enc > 3 ? P->n_enc : 3) * c.B < 6L * device_cus() && !P->seq_bf16) {        // (one-row BPTT tiles)
    for (int i = 0; i < P->n_enc + 3; ++i) {
      const int hh = i < P->n_enc ? P->enc[i].h : P->dec[i - P->n_enc].h;
      if (hh > MFM_SEQ_MAX_RESIDENT_H) continue;
      const int64_t HKB = round_up(4 * round_up(cdiv(hh, 4), 2), 16);
      P->wt_img[i] = carve(cur, 4 * HKB * HKB);
    }
  }
  if (V == 0 && c.B <= DWR_ROWS) {
    P->dw_flags = carve(cur, (int64_t)4 * P->T * DWR_ROWS + 4 * DWR_ROWS);
    P->dw_table = carve(cur, (int64_t)DWR_TABLE_CAP * 4);
  }
  P->lat_items_off = carve(cur, (int64_t)P->lat_items.size());
  P->lat_grd = carve(cur, (int64_t)c.B * rs);
  P->lat_rec = carve(cur, (int64_t)c.B * rs);
  P->lat_seed = (V == 2) ? carve(cur, (int64_t)c.B * rs) : -1;
  // MMD beyond the reference's batch size: scratch for the Gram-matrix form (mmd.hip).  Measured (MOSI sizes, us for the four
  // terms, row kernel vs GEMM form): B = 32: 18.8 vs 20.8, 64: 31 vs 21, 96: 44 vs 22, 128: 56 vs 22, 256: 105 vs 29, 512: 556 vs 50,
  // 1024: 1099 vs 117 -> from B = 48; MFM_MMD_GEMM_MINB moves the threshold (0 = never)
  {
    long minb = 48;
    if (const char* e = opt_get("MFM_MMD_GEMM_MINB")) minb = atol(e);
    P->mmd_scr = (V == 2 && minb > 0 && c.B >= minb && c.B <= 8192) ? carve(cur, mmd_scratch_floats(c.B, 4)) : -1;
  }
  if (V != 0) P->dh_last[3] = carve(cur, (int64_t)c.B * P->nzy);     // d loss / d [mu_y | logvar_y] (or z_y)
  P->yhat = carve(cur, (int64_t)c.B * c.output_dim);
  P->ones = carve(cur, TB);
  P->losses = carve(cur, 64);        // loss slots [MFM_LOSS_SLOTS], then the plan's device-side state (MfmPlan::ST_*)
  P->ws_floats = cur;
  return MFM_OK;
}

// Replay counters (mfm_plan_state_layout): device words that only a CAPTURED step advances -- one tick node behind the
// forward, one behind a backward with role workgroups -- so that every replay of a hipGraph draws new dropout masks and
// stamps its hand-over flags with an epoch of its own (the kernel arguments of a captured launch are frozen; the host part
// of both, the plan's call counter, is what eager calls advance).
__global__ void tick_kernel(unsigned long long* t64, unsigned* t32) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (t64) *t64 += 1ull;
    if (t32) *t32 += 1u;
  }
}
// non-role backward of a plan whose forward may have raised the status word: keep the step away from the parameters
__global__ void guard_propagate_kernel(const unsigned* status, float* guard) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && *status != 0u) *guard = __builtin_nanf("");
}
static bool stream_capturing(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
  return st == hipStreamCaptureStatusActive;
}
// host part of a hand-over epoch: consecutive launches of one plan -- eager calls and replays of graphs captured at
// different call counts, in any order -- must never carry the same value (epoch = this + replay counter, ho_epoch)
static unsigned epoch_base(uint64_t calls) { return (unsigned)calls * 0x9E3779B1u; }

struct Timer {
  MfmPlan* P; hipStream_t s; int kid; TimingPair* tp;
  Timer(MfmPlan* P_, hipStream_t s_, int kid_) : P(P_), s(s_), kid(kid_), tp(nullptr) {
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
  }
  ~Timer() { if (tp) (void)hipEventRecord(tp->b, s); }
};

#define RUN(kid, expr)                        \
  do {                                        \
    Timer _t(P, s, kid);                      \
    int _rc = (expr);                         \
    if (_rc != MFM_OK) return _rc;            \
  } while (0)

static MfmSeqDesc seq_desc(const MfmPlan* P, const SeqBuf& sb, int pbase, const float* params, float* W, bool dec) {
  MfmSeqDesc d;
  memset(&d, 0, sizeof(d));
  d.gates = W + sb.gates; d.hs = W + sb.hs; d.cs = W + sb.cs;
  d.w_ih = params + P->off[pbase + W_IH];
  d.w_hh = params + P->off[pbase + W_HH];
  d.b_ih = params + P->off[pbase + B_IH];
  d.b_hh = params + P->off[pbase + B_HH];
  d.h = sb.h; d.is_dec = dec ? 1 : 0;
  if (sb.wpack >= 0 && sb.h <= MFM_SEQ_MAX_RESIDENT_H) d.w_pack = W + sb.wpack;   // bf16 plans: fragments packed by K_PACK
  d.store_bf16 = P->st16 ? 1 : 0;
  d.bf16_dot = (P->cfg.precision && !P->seq_bf16 && P->opt_bf16_dot) ? 1 : 0;
  return d;
}

// element offset helpers
static inline const float* PW(const MfmPlan* P, const float* params, int idx) { return params + P->off[idx]; }

// ---- Memory Fusion Network, forward (reference mfm_model.py:140-199 restructured, see mfn_att.hip / mfn_mem.hip):
// cStar gather -> att1_fc1 (+relu/dropout) -> att1_fc2 -> softmax * cStar -> {att2_fc1 (+relu/dropout), attended part
// of gamma1_fc1 / gamma2_fc1} -> att2_fc2 (+tanh) -> memory recurrence -> heads on [h_l, h_a, h_v, mem]
// The MFN attention block as one launch per direction (mfn_att_fused.hip): OPT-IN with MFM_MFN_FUSED=1 (fp32 plans, T*B up
// to MFM_MFN_FUSED_MAXROWS, sizes that fit its LDS tiles).  Parity-tested on every MFN case, but measured slower than the
// GEMM launches it replaces (60 vs ~46 us forward at T*B = 640: profiles/r02_mfn_att_fused.txt), so the default stays off.
static bool mfn_fused_desc(const MfmPlan* P, const float* params, float* W, MfnAttFused& F) {
  const MfmPlanConfig& c = P->cfg;
  const PIdx& pi = P->pi;
  memset(&F, 0, sizeof(F));
  int off = 0;
  for (int m = 0; m < 3; ++m) {
    F.cs[m] = W + P->enc[3 + m].cs; F.dcx[m] = W + P->dcx[m];
    F.h[m] = P->enc[3 + m].h; F.Hp[m] = P->enc[3 + m].Hp; F.off[m] = off; off += F.h[m];
  }
  F.tot = P->tot; F.A2 = P->A2; F.T = P->T; F.B = P->B;
  F.nn1 = c.nn1; F.nn2 = c.nn2; F.g1 = c.g1; F.g2 = c.g2; F.M = c.mem_dim;
  F.w_att1_1 = PW(P, params, pi.att1_1); F.b_att1_1 = PW(P, params, pi.att1_1 + 1);
  F.w_att1_2 = PW(P, params, pi.att1_2); F.b_att1_2 = PW(P, params, pi.att1_2 + 1);
  F.w_att2_1 = PW(P, params, pi.att2_1); F.b_att2_1 = PW(P, params, pi.att2_1 + 1);
  F.w_att2_2 = PW(P, params, pi.att2_2); F.b_att2_2 = PW(P, params, pi.att2_2 + 1);
  F.w_gam1 = PW(P, params, pi.g1_1); F.b_gam1 = PW(P, params, pi.g1_1 + 1);
  F.w_gam2 = PW(P, params, pi.g2_1); F.b_gam2 = PW(P, params, pi.g2_1 + 1);
  F.cstar = W + P->cstar; F.h1 = W + P->h1; F.m1 = W + P->m1; F.att = W + P->att; F.attended = W + P->attended;
  F.h2 = W + P->h2; F.m2 = W + P->m2; F.a1 = W + P->a1; F.a2 = W + P->a2; F.chat = W + P->chat;
  F.dchat = W + P->dchat; F.du1 = W + P->du1; F.du2 = W + P->du2;
  F.dh2 = W + P->dh2; F.dlog = W + P->dlog; F.dh1 = W + P->dh1;
  F.p1 = c.drop_nn1; F.p2 = c.drop_nn2;
  if (c.precision != 0) return false;
  const char* on = opt_get("MFM_MFN_FUSED");
  if (!on || atoi(on) == 0) return false;
  long max_rows = 5120;
  if (const char* e = opt_get("MFM_MFN_FUSED_MAXROWS")) max_rows = atol(e);
  if ((int64_t)P->T * P->B > max_rows) return false;
  return mfn_att_fused_supported(F);
}

// The heads on [h_l, h_a, h_v](T-1) | mem_T folded into the memory-recurrence launches (fp32 plans; MFM_MFN_HEADS_FOLD=0:
// the grouped-GEMM form, which bf16 plans keep for their bf16-rounded operands)
static bool mfn_heads_desc(const MfmPlan* P, const float* params, float* W, MfnHeadsDev& H) {
  const MfmPlanConfig& c = P->cfg;
  const PIdx& pi = P->pi;
  memset(&H, 0, sizeof(H));
  H.tot = P->tot; H.nheads = (c.variant == 1) ? 2 : 1; H.zy = c.zy; H.nzy = P->nzy;
  for (int m = 0; m < 3; ++m) {
    const SeqBuf& sb = P->enc[3 + m];
    H.seg[m] = P->st16 ? W + P->h_last[3 + m] : W + sb.hs + (int64_t)(P->T - 1) * P->B * sb.Hp;
    H.seg_ld[m] = sb.Hp; H.seg_n[m] = sb.h;
  }
  H.w[0] = PW(P, params, pi.to_z[3]); H.b[0] = PW(P, params, pi.to_z[3] + 1);
  if (H.nheads == 2) { H.w[1] = PW(P, params, pi.to_lv[3]); H.b[1] = PW(P, params, pi.to_lv[3] + 1); }
  H.zyin = W + P->zyin; H.dz = W + P->dh_last[3]; H.d_hT = W + P->d_hT;
  bool on = c.precision == 0;
  if (const char* e = opt_get("MFM_MFN_HEADS_FOLD")) on = on && atoi(e) != 0;
  H.on = on ? 1 : 0;
  return on;
}

static int mfn_forward(MfmPlan* P, const float* params, int train, uint64_t seed, float* W, hipStream_t s) {
  const MfmPlanConfig& c = P->cfg;
  const PIdx& pi = P->pi;
  const int T = P->T, B = P->B, M = c.mem_dim, A2 = P->A2;
  const int64_t TB = (int64_t)T * B;
  const int prec = c.precision;
  GemmEpiSet es;
  memset(&es, 0, sizeof(es));
  es.seed = seed * 0x9E3779B97F4A7C15ull + P->calls * 0xD1B54A32D192ED03ull; es.train = train;
  es.tick = reinterpret_cast<const unsigned long long*>(P->tick_ptr(W));
  MfnAttFused F;
  if (mfn_fused_desc(P, params, W, F)) {
    F.train = train; F.seed = es.seed;
    RUN(K_MFN_ATT_FWD, mfn_att_fused_fwd_launch(F, s));
  } else {
    MfnCs cs;
    memset(&cs, 0, sizeof(cs));
    for (int m = 0; m < 3; ++m) { cs.cs[m] = W + P->enc[3 + m].cs; cs.h[m] = P->enc[3 + m].h; }
    cs.T = T; cs.B = B;
    RUN(K_MFN_GLUE, mfn_cstar_launch(cs, W + P->cstar, s));
    auto lin = [&](const float* a, int lda, int k, int widx, int n, float* cout, int ldc, int ldw) {
      MfmGemmDesc d;
      memset(&d, 0, sizeof(d));
      d.a = a; d.a_sm = lda; d.a_sk = 1;
      d.b = PW(P, params, widx); d.b_sn = ldw; d.b_sk = 1;
      d.c = cout; d.ldc = ldc; d.bias = PW(P, params, widx + 1);
      d.m = (int)TB; d.n = n; d.n_valid = n; d.k = k; d.batch = 1; d.split_k = 1; d.alpha = 1.0f;
      return d;
    };
    // fp32 plans with few rows (T*B <= 5120, the measured crossover): the four forward products as row-block launches
    // (lin_rows.hip: all operands of a workgroup requested at once, 9.2 instead of 13.7 us per launch at T*B = 640;
    // profiles/r02_lin_rows.txt); bf16 plans, larger batches and MFM_LIN_ROWS=0 keep the grouped GEMM
    long lr_max = 5120;
    if (const char* e = opt_get("MFM_LIN_ROWS_MAXROWS")) lr_max = atol(e);
    const bool lr_on = prec == 0 && TB <= lr_max && !(opt_get("MFM_LIN_ROWS") && atoi(opt_get("MFM_LIN_ROWS")) == 0);
    auto rows = [&](const MfmGemmDesc& d, int kind, float* aux, float p, unsigned op_id) {
      LinRowsItem it;
      memset(&it, 0, sizeof(it));
      it.a = d.a; it.lda = (int)d.a_sm; it.w = d.b; it.ldw = (int)d.b_sn; it.bias = d.bias; it.c = d.c; it.ldc = (int)d.ldc;
      it.n = d.n; it.k = d.k; it.kind = kind; it.aux = aux; it.p = p; it.op_id = op_id;
      return it;
    };
    {   // h1 = drop(relu(att1_fc1(cStar)))
      MfmGemmDesc g = lin(W + P->cstar, A2, A2, pi.att1_1, c.nn1, W + P->h1, c.nn1, A2);
      GemmEpi e = {W + P->m1, c.drop_nn1, 1, 101u, 0};
      LinRowsItem it = rows(g, 1, e.aux, e.p, e.op_id);
      es.epi = &e; es.count = 1;
      if (lr_on && lin_rows_supported(&it, 1, (int)TB)) RUN(K_MFN_ATT_FWD, lin_rows_launch(&it, 1, (int)TB, train, es.seed, s, es.tick));
      else RUN(K_MFN_ATT_FWD, gemm_group_launch(&g, 1, s, nullptr, nullptr, 0, prec, &es));
    }
    {   // logits = att1_fc2(h1)
      MfmGemmDesc g = lin(W + P->h1, c.nn1, c.nn1, pi.att1_2, A2, W + P->att, A2, c.nn1);
      LinRowsItem it = rows(g, 0, nullptr, 0.0f, 0u);
      if (lr_on && lin_rows_supported(&it, 1, (int)TB)) RUN(K_MFN_ATT_FWD, lin_rows_launch(&it, 1, (int)TB, train, es.seed, s, es.tick));
      else RUN(K_MFN_ATT_FWD, gemm_group_launch(&g, 1, s, nullptr, nullptr, 0, prec));
    }
    RUN(K_MFN_GLUE, mfn_softmax_fwd_launch(W + P->att, W + P->cstar, W + P->attended, TB, A2, s));
    {   // h2 = drop(relu(att2_fc1(attended))) ; a_n = gamma_n_fc1[:, :A2] attended + b   (the memory columns: mfn_mem)
      MfmGemmDesc g[3];
      g[0] = lin(W + P->attended, A2, A2, pi.att2_1, c.nn2, W + P->h2, c.nn2, A2);
      g[1] = lin(W + P->attended, A2, A2, pi.g1_1, c.g1, W + P->a1, c.g1, A2 + M);
      g[2] = lin(W + P->attended, A2, A2, pi.g2_1, c.g2, W + P->a2, c.g2, A2 + M);
      GemmEpi e = {W + P->m2, c.drop_nn2, 1, 102u, 0};
      LinRowsItem it[3] = {rows(g[0], 1, e.aux, e.p, e.op_id), rows(g[1], 0, nullptr, 0.0f, 0u), rows(g[2], 0, nullptr, 0.0f, 0u)};
      es.epi = &e; es.count = 1;
      if (lr_on && lin_rows_supported(it, 3, (int)TB)) RUN(K_MFN_ATT_FWD, lin_rows_launch(it, 3, (int)TB, train, es.seed, s, es.tick));
      else RUN(K_MFN_ATT_FWD, gemm_group_launch(g, 3, s, nullptr, nullptr, 0, prec, &es));
    }
    {   // cHat = tanh(att2_fc2(h2))
      MfmGemmDesc g = lin(W + P->h2, c.nn2, c.nn2, pi.att2_2, M, W + P->chat, M, c.nn2);
      GemmEpi e = {nullptr, 0.0f, 2, 0u, 0};
      LinRowsItem it = rows(g, 2, nullptr, 0.0f, 0u);
      es.epi = &e; es.count = 1;
      if (lr_on && lin_rows_supported(&it, 1, (int)TB)) RUN(K_MFN_ATT_FWD, lin_rows_launch(&it, 1, (int)TB, train, es.seed, s, es.tick));
      else RUN(K_MFN_ATT_FWD, gemm_group_launch(&g, 1, s, nullptr, nullptr, 0, prec, &es));
    }
  }
  {   // gamma gates + memory update for all T (mfm_model.py:177-181)
    MfmMemDesc md;
    memset(&md, 0, sizeof(md));
    md.a1 = W + P->a1; md.a2 = W + P->a2; md.chat = W + P->chat;
    md.w1m = PW(P, params, pi.g1_1) + A2; md.w2m = PW(P, params, pi.g2_1) + A2; md.ld_wm = A2 + M;
    md.w1b = PW(P, params, pi.g1_2); md.b1b = PW(P, params, pi.g1_2 + 1);
    md.w2b = PW(P, params, pi.g2_2); md.b2b = PW(P, params, pi.g2_2 + 1);
    md.gam1 = W + P->gam1; md.gam2 = W + P->gam2; md.mems = W + P->mems; md.mem_out = W + P->mem_out;
    md.T = T; md.B = B; md.M = M; md.H1 = c.g1; md.H2 = c.g2; md.train = train;
    md.p1 = c.drop_g1; md.p2 = c.drop_g2; md.seed = es.seed ^ 0x5DEECE66Dull;
    md.seed_dev = reinterpret_cast<const uint64_t*>(es.tick);
    MfnHeadsDev H;
    mfn_heads_desc(P, params, W, H);
    RUN(K_MFN_MEM_FWD, mfn_mem_fwd_launch(&md, &H, s));
    if (H.on) return MFM_OK;          // mu_y (and logvar_y) came out of the same launch
  }
  {   // heads on mfn_last = [h_l(T-1), h_a(T-1), h_v(T-1), mem]: mu_y (and logvar_y), summed over the four segments
    // into the zero-filled latent input (accumulating problems; the bias rides on the first segment)
    MfmGemmDesc g[8];
    int n = 0;
    const int nheads = (c.variant == 1) ? 2 : 1;
    for (int hd = 0; hd < nheads; ++hd) {
      const int widx = hd == 0 ? pi.to_z[3] : pi.to_lv[3];
      int koff = 0;
      for (int sg = 0; sg < 4; ++sg) {
        const SeqBuf* sb = sg < 3 ? &P->enc[3 + sg] : nullptr;
        MfmGemmDesc d;
        memset(&d, 0, sizeof(d));
        d.a = sb ? (P->st16 ? W + P->h_last[3 + sg] : W + sb->hs + (int64_t)(T - 1) * B * sb->Hp) : W + P->mem_out;
        d.a_sm = sb ? sb->Hp : M; d.a_sk = 1;
        const int k = sb ? sb->h : M;
        d.b = PW(P, params, widx) + koff; d.b_sn = P->tot + M; d.b_sk = 1;
        d.c = W + P->zyin + hd * c.zy; d.ldc = P->nzy;
        if (sg == 0) d.bias = PW(P, params, widx + 1);
        d.m = B; d.n = c.zy; d.n_valid = c.zy; d.k = k; d.batch = 1; d.split_k = 1; d.accumulate = 1; d.alpha = 1.0f;
        g[n++] = d;
        koff += k;
      }
    }
    RUN(K_MFN_HEADS, gemm_group_launch(g, n, s, nullptr, nullptr, 0, prec));
  }
  return MFM_OK;
}

static int forward(MfmPlan* P, const float* params, const float* x, const void* y, int train, uint64_t seed,
                   float* W, float* xhat_out[3], float* yhat_out, float* losses_out, hipStream_t s,
                   float* grads_to_zero = nullptr) {
  OptScope _opts(P->opts);         // every MFM_* switch below this call: the plan's table, not the environment
  const MfmPlanConfig& c = P->cfg;
  const PIdx& pi = P->pi;
  const int V = c.variant;
  const int T = P->T, B = P->B;
  const int64_t TB = (int64_t)T * B;
  float* losses = losses_out ? losses_out : W + P->losses;
  if (V == 2) MFM_REQUIRE(P->gauss, "plan (MFM / MMD variant): call mfm_plan_set_gauss before the forward");
  // the loss slots (and, in the fused step, the gradient buffer; variants 1, 2: the MFN's accumulation targets)
  // are cleared by the first GEMM launch
  ZeroSpans zs;
  memset(&zs, 0, sizeof(zs));
  if ((((uintptr_t)losses) & 15) == 0) { zs.ptr[0] = losses; zs.n[0] = MFM_LOSS_SLOTS; }
  else MFM_HIP_CHECK(hipMemsetAsync(losses, 0, MFM_LOSS_SLOTS * sizeof(float), s));
  P->grads_prezeroed = nullptr;
  if (grads_to_zero && (((uintptr_t)grads_to_zero) & 15) == 0 && (P->n_params & 3) == 0) {
    zs.ptr[1] = grads_to_zero; zs.n[1] = P->n_params;
    P->grads_prezeroed = grads_to_zero;
  }
  if (V != 0) { zs.ptr[2] = W + P->zero_blk; zs.n[2] = P->zero_len; }
  // up to 5120 rows: decoder fc1, the squared error AND (training) dH = dx_hat Wfc run as one launch (dec_fc1.hip) whose
  // column groups add into dH (bf16 plans: operands rounded to bf16 in the kernel); larger T*B, shapes it does not take and
  // MFM_FC1_FUSED=0 use the grouped GEMMs (F4, B0)
  const bool fc1_env_on = !(opt_get("MFM_FC1_FUSED") && atoi(opt_get("MFM_FC1_FUSED")) == 0);
  long fc1_max_rows = 5120;                        // measured crossover (profiles/r02_dec_fc1.txt)
  if (const char* e = opt_get("MFM_FC1_FUSED_MAXROWS")) fc1_max_rows = atol(e);
  const bool fc1_fused = fc1_env_on && TB <= fc1_max_rows && !P->st16;      // (the fused kernel reads fp32 hidden states)
  if (fc1_fused && train) { zs.ptr[3] = W + P->dhs_blk; zs.n[3] = P->dhs_len; }
  P->calls++;

  // bf16 plans: the recurrences' weight fragments, rounded and packed once per step (lstm_seq_bf16.hip)
  const bool seq_bf16 = P->seq_bf16;
  const bool st16 = P->st16;
  if (seq_bf16) {
    MfmSeqDesc q[9];
    int n = 0;
    for (int e = 0; e < P->n_enc; ++e) q[n++] = seq_desc(P, P->enc[e], P->enc_p[e], params, W, false);
    for (int m = 0; m < 3; ++m) q[n++] = seq_desc(P, P->dec[m], P->dec_p[m], params, W, true);
    // ONE launch for every weight image of the step (pack_dev.h): the recurrences' fragments, and on bf16-resident plans the
    // projection tiles + biases and the decoders' fc1 images
    PackLaunch PKL;
    { const int rc0 = lstm_pack_prepare(q, n, &PKL); if (rc0 != MFM_OK) return rc0; }
    PjPackDev PJD;
    Fc1PackArgs FCA;
    const PjPackDev* pjp = nullptr;
    const Fc1PackArgs* fcp = nullptr;
    if (st16 && P->proj16) {
      PanelLaunch PL;
      memset(&PL, 0, sizeof(PL));
      PL.ngroups = P->n_enc;
      for (int e = 0; e < P->n_enc; ++e) {
        const SeqBuf& sb = P->enc[e];
        const int pb = P->enc_p[e];
        PanelGroup& G = PL.g[e];
        G.w = params + P->off[pb + W_IH]; G.ldw = P->enc_d[e];
        G.bias = params + P->off[pb + B_IH]; G.bias2 = params + P->off[pb + B_HH];
        G.n = 4 * sb.Hp; G.seg = sb.Hp; G.seg_valid = sb.h; G.k_off = P->enc_xoff[e]; G.k_len = P->enc_d[e];
      }
      const int rc0 = proj_pack_prepare(PL, P->pj, W + P->pj_wimg, W + P->pj_bimg, &PJD);
      if (rc0 != MFM_OK) return rc0;
      pjp = &PJD; P->pj_pack_call = P->calls;
    }
    if (st16 && train && !(xhat_out && (xhat_out[0] || xhat_out[1] || xhat_out[2])) &&
        !(opt_get("MFM_FC1_LARGE") && atoi(opt_get("MFM_FC1_LARGE")) == 0)) {
      DecFc1LargeLaunch FLp;
      memset(&FLp, 0, sizeof(FLp));
      FLp.n_items = 3; FLp.rows = (int)TB;
      for (int m = 0; m < 3; ++m) {
        DecFc1LargeItem& I = FLp.it[m];
        I.w = params + P->off[P->dec_p[m] + FC_W]; I.d = P->dec_d[m]; I.h = P->dec[m].h; I.Hp = P->dec[m].Hp;
        I.ld_dxhat = P->dxh_ld[m]; I.ldx = P->D; I.wimg = W + P->fc1_wimg[m];
      }
      if (dec_fc1_large_uses_wimg(FLp)) {
        const int rc0 = fc1_pack_prepare(FLp, &FCA);
        if (rc0 != MFM_OK) return rc0;
        fcp = &FCA; P->fc1_pack_call = P->calls;
      }
    }
    RUN(K_PACK, pack_all_launch(&PKL, pjp, fcp, s));
  }

  // F0: input projections.  MFM_KL_EF at B <= 32 (fp32 plans on the fold launches): produced
  // by role workgroups of the encoder launch itself (proj_role_dev.h), which also clear the zero spans
  bool proj_in_fold = false;
  const bool capturing = stream_capturing(s);
  if (V == 0 && !seq_bf16 && !st16 && P->n_enc == 4 && P->fold_state >= 0 && P->projfold_state >= 0 && P->pf_flags >= 0 &&
      P->opt_handover && TB * P->D < ((int64_t)1 << 28)) {
    int hh[4], kk[4];
    for (int e = 0; e < 4; ++e) { hh[e] = P->enc[e].h; kk[e] = P->enc_d[e]; }
    proj_in_fold = seq_small_foldproj_supported(T, B, hh, kk, 4);
    if (!proj_in_fold) P->projfold_state = -1;
  }
  auto run_f0 = [&]() -> int {
    MfmGemmDesc g[6];
    memset(g, 0, sizeof(g));
    for (int e = 0; e < P->n_enc; ++e) {
      const SeqBuf& sb = P->enc[e];
      const int pb = P->enc_p[e];
      MfmGemmDesc& d = g[e];
      d.a = x + P->enc_xoff[e]; d.a_sm = P->D; d.a_sk = 1; d.a_sz = 0;
      d.b = params + P->off[pb + W_IH]; d.b_sz = (int64_t)sb.h * P->enc_d[e]; d.b_sn = P->enc_d[e]; d.b_sk = 1;
      d.c = W + sb.gates; d.c_sz = sb.Hp; d.ldc = 4 * (int64_t)sb.Hp;
      d.c_bf16 = st16 ? 1 : 0;                        // bf16-resident x-projection (same element offsets)
      d.bias = params + P->off[pb + B_IH]; d.bias2 = params + P->off[pb + B_HH]; d.bias_sz = sb.h;
      d.m = (int)TB; d.n = sb.Hp; d.n_valid = sb.h; d.k = P->enc_d[e]; d.batch = 4; d.split_k = 1;
      d.alpha = 1.0f;
    }
    // large batches: the row-panel kernel reads x once for all encoders (gemm_panel.hip).  Its launcher picks the panel
    // height and declines when its cost model favours the tiled kernel (measured crossover, profiles/r02_gemm_panel.txt:
    // T*B ~ 10240 in both dtypes at the MOSI sizes -- equal at B = 512, panel 135 vs 166 us fp32 and 86 vs 104 us bf16 at
    // B = 640); MFM_PANEL_MINROWS=n forces the panel kernel from n rows on (and the tiled one below)
    const char* pe = opt_get("MFM_PANEL_MINROWS");
    const bool panel_forced = pe && TB >= atol(pe);
    const bool panel = (pe ? panel_forced : TB >= 16L * device_cus()) && P->n_enc <= MFM_PANEL_MAXG && (int64_t)TB * P->D < ((int64_t)1 << 29);
    if (panel || (st16 && P->proj16)) {
      PanelLaunch PL;
      memset(&PL, 0, sizeof(PL));
      PL.a = x; PL.lda = P->D; PL.M = (int)TB; PL.K = P->D;
      for (int e = 0; e < P->n_enc; ++e) {
        const SeqBuf& sb = P->enc[e];
        const int pb = P->enc_p[e];
        PanelGroup& G = PL.g[PL.ngroups++];
        G.w = params + P->off[pb + W_IH]; G.ldw = P->enc_d[e];
        G.bias = params + P->off[pb + B_IH]; G.bias2 = params + P->off[pb + B_HH];
        G.c = W + sb.gates; G.ldc = 4 * (int64_t)sb.Hp; G.c_bf16 = st16 ? 1 : 0;
        G.n = 4 * sb.Hp; G.seg = sb.Hp; G.seg_valid = sb.h; G.k_off = P->enc_xoff[e]; G.k_len = P->enc_d[e];
      }
      if (st16 && P->proj16) {
        const int src0[3] = {0, c.d_l, c.d_l + c.d_a}, nn[3] = {c.d_l, c.d_a, c.d_v};
        if (P->pj_pack_call != P->calls) RUN(K_PACK, proj_bf16_pack_launch(PL, P->pj, W + P->pj_wimg, W + P->pj_bimg, s));
        RUN(K_PROJ, proj_bf16_launch(PL, P->pj, W + P->pj_wimg, W + P->pj_bimg, W + P->x16, P->x16_ld, src0, nn, P->x16_off, &zs, s));
        P->x16_call = P->calls;
      } else if (gemm_panel_pays(PL, c.precision, panel_forced)) RUN(K_PROJ, gemm_panel_launch(PL, &zs, c.precision, panel_forced, s));
      else RUN(K_PROJ, gemm_group_launch(g, P->n_enc, s, &zs, nullptr, 0, c.precision));
    } else {
      RUN(K_PROJ, gemm_group_launch(g, P->n_enc, s, &zs, nullptr, 0, c.precision));
    }
    return MFM_OK;
  };
  if (!proj_in_fold) { const int rc0 = run_f0(); if (rc0 != MFM_OK) return rc0; }
  // training steps: this step's transposed-weight images for the one-row BPTT kernels (lstm_seq_dev.h), written by idle
  // workgroups of the encoder recurrence launch
  WtImgItem wt_items[MFM_WT_MAX];
  int n_wt_items = 0;
  if (train && !seq_bf16 && !(opt_get("MFM_WT_IMG") && atoi(opt_get("MFM_WT_IMG")) == 0)) {
    bool all = true;
    for (int i = 0; i < P->n_enc + 3; ++i) all = all && P->wt_img[i] >= 0;
    if (all) {
      n_wt_items = P->n_enc + 3;
      for (int i = 0; i < n_wt_items; ++i) {
        const bool dec = i >= P->n_enc;
        const SeqBuf& sb = dec ? P->dec[i - P->n_enc] : P->enc[i];
        const int pb = dec ? P->dec_p[i - P->n_enc] : P->enc_p[i];
        WtImgItem& I = wt_items[i];
        I.w_hh = params + P->off[pb + W_HH];
        I.w_ih = dec ? params + P->off[pb + W_IH] : nullptr;       // decoders, steps >= 1: W_ih + W_hh (mfm_model.py:85)
        I.img = W + P->wt_img[i]; I.h = sb.h;
        I.HKB = round_up(4 * round_up(cdiv(sb.h, 4), 2), 16);
      }
    }
  }
  // the latent stack's launch descriptor (used by F2, or by the fold launch of F1)
  LatentDev L = P->lat;
  {
    L.ops = reinterpret_cast<const LatOp*>(W + P->lat_ops_off);
    L.items_fwd = reinterpret_cast<const int*>(W + P->lat_items_off);
    L.items_bwd = L.items_fwd + (size_t)4 * MFM_LAT_MAXSTAGES * MFM_LAT_ROW_THREADS * 4;
    if (opt_get("MFM_LATENT_DBG")) L.dbg = reinterpret_cast<unsigned long long*>(W + P->dbg_off);
    for (int e = 0; e < 4; ++e) {
      if (e == 3 && V != 0) { L.enc_h[e] = W + P->zyin; L.enc_ld[e] = P->nzy; continue; }
      L.enc_h[e] = st16 ? W + P->h_last[e] : W + P->enc[e].hs + (int64_t)(T - 1) * B * P->enc[e].Hp;
      L.enc_ld[e] = P->enc[e].Hp;
    }
    for (int m = 0; m < 3; ++m) { L.dec_init[m] = W + P->dec_init[m]; L.dec_ld[m] = P->dec_h[m]; }
    L.rec = W + P->lat_rec;
    L.yhat_out = yhat_out ? yhat_out : W + P->yhat;
    L.y = y; L.losses = losses; L.train = train;
    // (+ the replay counter, added by the kernels; the large odd stride keeps eager calls and replays of captured steps on
    // distinct streams)
    L.seed = seed * 0x9E3779B97F4A7C15ull + P->calls * 0xD1B54A32D192ED03ull;
    L.tick = reinterpret_cast<const unsigned long long*>(P->tick_ptr(W));
  }
  // F1: encoder recurrences (up to MFM_MAX_SEQ per launch).  MFM_KL_EF at small batches: the four encoders' workgroups
  // also run their rows' latent chains (fold launch, lstm_seq_small.hip) and F2 disappears
  bool folded = false;
  if (proj_in_fold) {
    MfmSeqDesc q[4];
    for (int e = 0; e < 4; ++e) q[e] = seq_desc(P, P->enc[e], P->enc_p[e], params, W, false);
    ProjRole PR;
    memset(&PR, 0, sizeof(PR));
    PR.x = x; PR.ldx = P->D; PR.x_rows = (int)TB;
    PR.flags = reinterpret_cast<unsigned*>(W + P->pf_flags); PR.epoch = epoch_base(P->calls); PR.tick = P->tick_ptr(W);
    // a consumer that gives up: status bit 0, NaN into the regulariser slot (what the module path returns as `kld`)
    PR.ctl = P->ho_ctl(W, losses + 4, 1u);
    PR.fault = (P->opt_fault == 1) ? 1 : 0;
    if (PR.fault) P->opt_fault = 0;
    PR.zs = zs;
    PR.loss_ptr = zs.ptr[0]; PR.loss_n = (int)zs.n[0];
    PR.bf16 = c.precision ? 1 : 0;
    // training steps: the BPTT launches of this step take their transposed weights from images the role workgroups write
    if (n_wt_items == 7) { for (int i = 0; i < 7; ++i) PR.wt[i] = wt_items[i]; PR.n_wt = 7; }
    PR.zs.ptr[0] = nullptr; PR.zs.n[0] = 0;
    for (int e = 0; e < 4; ++e) {
      const int pb = P->enc_p[e];
      PR.e[e].w = params + P->off[pb + W_IH]; PR.e[e].b_ih = params + P->off[pb + B_IH]; PR.e[e].b_hh = params + P->off[pb + B_HH];
      PR.e[e].k_off = P->enc_xoff[e]; PR.e[e].k = P->enc_d[e];
    }
    int rc;
    { Timer _t(P, s, K_ENC_FWD); rc = seq_foldproj_launch(q, 4, T, B, L, params, PR, s); }
    if (rc == MFM_OK) { folded = true; P->projfold_state = 1; P->fold_state = 1; P->ever_handover = true; if (PR.n_wt) P->wt_call = P->calls; }
    else if (rc == MFM_ERR_UNSUPPORTED) {
      P->projfold_state = -1;
      const int rc0 = run_f0();
      if (rc0 != MFM_OK) return rc0;
    } else return rc;
  }
  if (!folded && V == 0 && !seq_bf16 && P->n_enc == 4 && P->fold_state >= 0) {
    MfmSeqDesc q[4];
    for (int e = 0; e < 4; ++e) q[e] = seq_desc(P, P->enc[e], P->enc_p[e], params, W, false);
    int rc;
    bool wrote = false;
    if (P->fold_state == 1) { Timer _t(P, s, K_ENC_FWD); rc = seq_fold_launch(q, 4, T, B, false, L, params, nullptr, s, nullptr, n_wt_items ? wt_items : nullptr, n_wt_items, &wrote); }
    else rc = seq_fold_launch(q, 4, T, B, false, L, params, nullptr, s, nullptr, n_wt_items ? wt_items : nullptr, n_wt_items, &wrote);
    if (rc == MFM_OK) { folded = true; P->fold_state = 1; if (wrote) P->wt_call = P->calls; }
    else if (rc == MFM_ERR_UNSUPPORTED) P->fold_state = (P->fold_state == 0) ? -1 : P->fold_state;
    else return rc;
  }
  for (int e0 = 0; e0 < P->n_enc && !folded; e0 += MFM_MAX_SEQ) {
    MfmSeqDesc q[MFM_MAX_SEQ];
    const int n = std::min(MFM_MAX_SEQ, P->n_enc - e0);
    for (int e = 0; e < n; ++e) {
      q[e] = seq_desc(P, P->enc[e0 + e], P->enc_p[e0 + e], params, W, false);
      if (st16) q[e].h_last = W + P->h_last[e0 + e];
    }
    if (!seq_bf16 && n_wt_items && e0 == 0 && n == P->n_enc) {
      bool wrote = false;
      RUN(K_ENC_FWD, seq_fwd_img_launch(q, n, T, B, wt_items, n_wt_items, &wrote, s));
      if (wrote) P->wt_call = P->calls;
    } else RUN(K_ENC_FWD, seq_bf16 ? mfm_lstm_seq_fwd_bf16(q, n, T, B, s) : mfm_lstm_seq_fwd(q, n, T, B, s));
  }
  if (V != 0) {
    int rc = mfn_forward(P, params, train, seed, W, s);
    if (rc != MFM_OK) return rc;
  }
  // F2: latent stack
  if (!folded) {
    RUN(K_LAT_FWD, latent_fwd_launch(L, params, s));
  }
  // MMD regulariser of the non-KL MFM on z_l, z_a, z_v, z_y (mfm_model.py:540-541): value into the reg slot, its
  // gradient (unscaled: the latent backward weighs it with lda_mmd or the upstream gradient) into its seed record
  if (V == 2) {
    const int rs = P->lat.rec_size;
    const int zn[4] = {c.zl, c.za, c.zv, c.zy};
    const int gl = c.zl + c.za + c.zv + c.zy;
    int goff = 0;
    MmdItem it[4];
    for (int e = 0; e < 4; ++e) {
      it[e].z = W + P->lat_rec + P->z_seg[e]; it[e].g = P->gauss + goff; it[e].dz = W + P->lat_seed + P->z_seg[e]; it[e].dim = zn[e];
      goff += zn[e];
    }
    RUN(K_MMD, mmd_group_launch(it, 4, rs, gl, rs, B, losses + 4, 1.0f, s, P->mmd_scr >= 0 ? W + P->mmd_scr : nullptr));     // the four terms in one launch (large B: three)
  }
  // F3: decoder recurrences
  {
    MfmSeqDesc q[3];
    for (int m = 0; m < 3; ++m) {
      q[m] = seq_desc(P, P->dec[m], P->dec_p[m], params, W, true);
      q[m].h_init = W + P->dec_init[m]; q[m].ld_init = P->dec_h[m];
    }
    RUN(K_DEC_FWD, seq_bf16 ? mfm_lstm_seq_fwd_bf16(q, 3, T, B, s) : mfm_lstm_seq_fwd(q, 3, T, B, s));
  }
  // F4: decoder fc1 -> x_hat
  float* xh[3];
  {
    MfmGemmDesc g[3];
    memset(g, 0, sizeof(g));
    for (int m = 0; m < 3; ++m) {
      const SeqBuf& sb = P->dec[m];
      const int pb = P->dec_p[m];
      xh[m] = (xhat_out && xhat_out[m]) ? xhat_out[m] : W + P->xhat[m];
      MfmGemmDesc& d = g[m];
      d.a = W + sb.hs; d.a_sm = sb.Hp; d.a_sk = 1; d.a_bf16 = st16 ? 1 : 0;
      d.b = params + P->off[pb + FC_W]; d.b_sn = sb.h; d.b_sk = 1;
      d.c = xh[m]; d.ldc = P->dec_d[m];
      // bf16-resident training steps need the squared error and d x_hat only: x_hat itself (53 MB at B=2048) is not written
      if (st16 && train && !(xhat_out && xhat_out[m])) d.c = nullptr;
      d.bias = params + P->off[pb + FC_B];
      d.m = (int)TB; d.n = P->dec_d[m]; d.n_valid = d.n; d.k = sb.h; d.batch = 1; d.split_k = 1; d.alpha = 1.0f;
    }
    // reconstruction losses + d x_hat in the same tiles (F5 of the first versions was its own launch)
    const float lda[3] = {c.lda_xl, c.lda_xa, c.lda_xv};
    MseEpi me[3];
    memset(me, 0, sizeof(me));
    for (int m = 0; m < 3; ++m) {
      const double cnt = (double)TB * P->dec_d[m];
      me[m].x = x + P->dec_xoff[m]; me[m].ldx = P->D;
      me[m].dxhat = W + P->dxhat[m];
      me[m].loss = losses + 1 + m;
      me[m].inv_count = (float)(1.0 / cnt);
      me[m].grad_scale = (float)(2.0 * lda[m] / cnt);
      if (st16) { me[m].dxhat_bf16 = 1; me[m].ld_dxhat = P->dxh_ld[m]; }
    }
    int rc = MFM_ERR_UNSUPPORTED;
    // bf16-resident training steps: fc1, the squared error, d x_hat and dH in one launch of persistent workgroups
    // (dec_fc1_large.hip); MFM_FC1_LARGE=0 keeps the two grouped GEMMs
    if (st16 && train && !(xhat_out && (xhat_out[0] || xhat_out[1] || xhat_out[2])) &&
        !(opt_get("MFM_FC1_LARGE") && atoi(opt_get("MFM_FC1_LARGE")) == 0)) {
      DecFc1LargeLaunch FL;
      memset(&FL, 0, sizeof(FL));
      FL.n_items = 3; FL.rows = (int)TB;
      bool ok = true;
      for (int m = 0; m < 3; ++m) {
        DecFc1LargeItem& I = FL.it[m];
        I.hs = g[m].a; I.w = g[m].b; I.bias = g[m].bias; I.x = me[m].x; I.ldx = me[m].ldx;
        I.dxhat = me[m].dxhat; I.ld_dxhat = P->dxh_ld[m]; I.dhs = W + P->dec_dhs[m]; I.loss = me[m].loss;
        I.d = P->dec_d[m]; I.h = P->dec[m].h; I.Hp = P->dec[m].Hp;
        I.inv_count = me[m].inv_count; I.grad_scale = me[m].grad_scale;
        I.wimg = W + P->fc1_wimg[m];
        ok = ok && dec_fc1_large_supported(I);
        FL.packed = (P->fc1_pack_call == P->calls) ? 1 : 0;
      }
      if (ok) {
        { Timer _t(P, s, K_FC1_FWD); rc = dec_fc1_large_launch(FL, s); }
        if (rc != MFM_OK) return rc;
        P->fc1_bwd_call = P->calls;               // dH is done: the backward skips its fc1 GEMM
      }
    }
    if (fc1_fused) {
      DecFc1Launch FL;
      memset(&FL, 0, sizeof(FL));
      FL.n_items = 3; FL.rows = (int)TB; FL.with_bwd = train ? 1 : 0; FL.bf16 = c.precision;
      for (int m = 0; m < 3; ++m) {
        DecFc1Item& I = FL.it[m];
        I.hs = g[m].a; I.w = g[m].b; I.bias = g[m].bias; I.x = me[m].x; I.ldx = me[m].ldx;
        I.xhat = xh[m]; I.dxhat = me[m].dxhat; I.dhs = W + P->dec_dhs[m]; I.loss = me[m].loss;
        I.d = P->dec_d[m]; I.h = P->dec[m].h; I.Hp = P->dec[m].Hp;
        I.inv_count = me[m].inv_count; I.grad_scale = me[m].grad_scale;
      }
      { Timer _t(P, s, K_FC1_FWD); rc = dec_fc1_launch(FL, train != 0, s); }
      if (rc == MFM_OK && train) P->fc1_bwd_call = P->calls;
      else if (rc != MFM_OK && rc != MFM_ERR_UNSUPPORTED) return rc;
    }
    if (rc == MFM_ERR_UNSUPPORTED) RUN(K_FC1_FWD, gemm_group_launch(g, 3, s, nullptr, me, 3, c.precision));
  }
  (void)pi;
  // captured into a hipGraph: every replay advances the device half of the call counter (dropout streams, hand-over epochs)
  if (capturing) {
    hipLaunchKernelGGL(tick_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<unsigned long long*>(P->tick_ptr(W)), (unsigned*)nullptr);
    MFM_LAUNCH_CHECK("tick_kernel");
  }
  return MFM_OK;
}

// `only_init`: just the decoders' t = 0 input product (the rest went to the one-pass kernel, dw_onepass.hip)
static void dA_gemms(const MfmPlan* P, const SeqBuf& sb, int pb, float* W, float* grads, std::vector<MfmGemmDesc>& out,
                     const float* xin, int64_t ldx, int kin, bool dec, bool only_init = false) {
  const int T = P->T, B = P->B;
  const int64_t TB = (int64_t)T * B;
  MfmGemmDesc base;
  memset(&base, 0, sizeof(base));
  base.a_sz = sb.Hp; base.a_sm = 1; base.a_sk = 4 * (int64_t)sb.Hp;
  base.m = sb.h; base.batch = 4; base.accumulate = 1; base.split_k = 0; base.alpha = 1.0f;
  base.a_bf16 = P->st16 ? 1 : 0;      // (bf16-resident plans come here for the decoders' t = 0 product only)
  // recurrent product sum_{t>=1} dA_t^T h_{t-1}
  if (T > 1 && !only_init) {
    MfmGemmDesc d = base;
    d.a = W + sb.gates + (int64_t)B * 4 * sb.Hp;
    d.b = W + sb.hs; d.b_sk = sb.Hp; d.b_sn = 1;
    d.k = (int)(TB - B); d.n = sb.h; d.n_valid = sb.h;
    d.c = grads + P->off[pb + W_HH]; d.c_sz = (int64_t)sb.h * sb.h; d.ldc = sb.h;
    if (dec) d.c2 = grads + P->off[pb + W_IH];   // steps >=1 feed h back as the input (mfm_model.py:85)
    out.push_back(d);
  }
  // input product: encoders sum_t dA_t^T x_t ; decoders dA_0^T h_init
  {
    MfmGemmDesc d = base;
    d.a = W + sb.gates;
    d.b = xin; d.b_sk = ldx; d.b_sn = 1;
    d.k = dec ? B : (int)TB; d.n = kin; d.n_valid = kin;
    d.c = grads + P->off[pb + W_IH]; d.c_sz = (int64_t)sb.h * kin; d.ldc = kin;
    out.push_back(d);
  }
  // biases: column sums of dA (both b_ih and b_hh)
  if (!only_init) {
    MfmGemmDesc d = base;
    d.a = W + sb.gates;
    d.b = W + P->ones; d.b_sk = 1; d.b_sn = 1;
    d.k = (int)TB; d.n = 1; d.n_valid = 1;
    d.c = grads + P->off[pb + B_IH]; d.c_sz = sb.h; d.ldc = 1;
    d.c2 = grads + P->off[pb + B_HH];
    out.push_back(d);
  }
}

// weights of the loss terms for a backward of  disc * L_disc + gen * sum_m lda_m MSE_m + reg * REG  (the module path's lazy
// losses, mfm_plan_backward_weighted): gen is a switch (the forward baked lda_m into d x_hat), disc and reg are factors
struct LossW { float disc; int gen_on; float reg; int write_disc; };

struct ExtGrads {           // upstream gradients supplied by the caller (autograd module path)
  const float* d_xhat[3];
  const float* d_yhat;
  const float* d_reg;       // device scalar
};

// ---- Memory Fusion Network, backward.  Appends the MFN's weight-gradient products to `tail`.
static int mfn_backward(MfmPlan* P, const float* params, float* W, float* grads, hipStream_t s,
                        std::vector<MfmGemmDesc>& tail) {
  const MfmPlanConfig& c = P->cfg;
  const PIdx& pi = P->pi;
  const int T = P->T, B = P->B, M = c.mem_dim, A2 = P->A2, tot = P->tot;
  const int64_t TB = (int64_t)T * B;
  const int prec = c.precision;
  // TN product for a weight gradient: C[m][n] (+)= sum_r A[r][m] B[r][n] over `rows` rows
  auto tn = [&](const float* a, int lda, int m, const float* b, int ldb, int n, float* cc, int ldc, int64_t rows) {
    MfmGemmDesc d;
    memset(&d, 0, sizeof(d));
    d.alpha = 1.0f; d.batch = 1; d.accumulate = 1; d.split_k = 0;
    d.a = a; d.a_sm = 1; d.a_sk = lda;
    d.b = b; d.b_sk = ldb; d.b_sn = 1;
    d.c = cc; d.ldc = ldc; d.m = m; d.n = n; d.n_valid = n; d.k = (int)rows;
    tail.push_back(d);
  };
  auto colsum = [&](const float* a, int lda, int m, float* cc, int64_t rows) { tn(a, lda, m, W + P->ones, 1, 1, cc, 1, rows); };
  // NN product: C[r][n] = sum_k A[r][k] Wt[k][n], Wt = a [K, N] row-major block with row stride ldw
  auto nn = [&](const float* a, int lda, int k, const float* wt, int ldw, int n, float* cc, int ldc, int64_t rows, int acc) {
    MfmGemmDesc d;
    memset(&d, 0, sizeof(d));
    d.alpha = 1.0f; d.batch = 1; d.split_k = 1; d.accumulate = acc;
    d.a = a; d.a_sm = lda; d.a_sk = 1;
    d.b = wt; d.b_sk = ldw; d.b_sn = 1;
    d.c = cc; d.ldc = ldc; d.m = (int)rows; d.n = n; d.n_valid = n; d.k = k;
    return d;
  };
  GemmEpiSet es;
  memset(&es, 0, sizeof(es));
  const int nheads = (c.variant == 1) ? 2 : 1;
  {   // through the heads on mfn_last: d h_T of the three MFN LSTMs and d mem_T (accumulated over the heads)
    MfmGemmDesc g[4];
    int n = 0;
    for (int hd = 0; hd < nheads; ++hd) {
      const int widx = hd == 0 ? pi.to_z[3] : pi.to_lv[3];
      const float* dz = W + P->dh_last[3] + hd * c.zy;
      g[n++] = nn(dz, P->nzy, c.zy, PW(P, params, widx), tot + M, tot, W + P->d_hT, tot, B, 1);
      g[n++] = nn(dz, P->nzy, c.zy, PW(P, params, widx) + tot, tot + M, M, W + P->dmem, M, B, 1);
      // dW_head[:, segment] = dz^T segment ; db = column sums of dz
      int koff = 0;
      for (int sg = 0; sg < 4; ++sg) {
        const SeqBuf* sb = sg < 3 ? &P->enc[3 + sg] : nullptr;
        const float* seg = sb ? (P->st16 ? W + P->h_last[3 + sg] : W + sb->hs + (int64_t)(T - 1) * B * sb->Hp) : W + P->mem_out;
        const int k = sb ? sb->h : M;
        tn(dz, P->nzy, c.zy, seg, sb ? sb->Hp : M, k, grads + P->off[widx] + koff, tot + M, B);
        koff += k;
      }
      colsum(dz, P->nzy, c.zy, grads + P->off[widx + 1], B);
    }
    MfnHeadsDev Hc;
    if (!mfn_heads_desc(P, params, W, Hc)) RUN(K_MFN_HEADS, gemm_group_launch(g, n, s, nullptr, nullptr, 0, prec));
  }
  {   // memory recurrence BPTT: dz_n (in gam_n), du_n, d(pre-tanh cHat)
    MfmMemDesc md;
    memset(&md, 0, sizeof(md));
    md.a1 = W + P->a1; md.a2 = W + P->a2; md.chat = W + P->chat;
    md.w1m = PW(P, params, pi.g1_1) + A2; md.w2m = PW(P, params, pi.g2_1) + A2; md.ld_wm = A2 + M;
    md.w1b = PW(P, params, pi.g1_2); md.b1b = PW(P, params, pi.g1_2 + 1);
    md.w2b = PW(P, params, pi.g2_2); md.b2b = PW(P, params, pi.g2_2 + 1);
    md.gam1 = W + P->gam1; md.gam2 = W + P->gam2; md.mems = W + P->mems;
    md.dmem_out = W + P->dmem; md.du1 = W + P->du1; md.du2 = W + P->du2; md.dchat = W + P->dchat;
    md.dchat_pre_tanh = 1;
    md.T = T; md.B = B; md.M = M; md.H1 = c.g1; md.H2 = c.g2; md.train = 1;
    md.p1 = c.drop_g1; md.p2 = c.drop_g2;
    MfnHeadsDev H;
    mfn_heads_desc(P, params, W, H);          // folded: d mem_T and d h_T are formed at the head of this launch
    RUN(K_MFN_MEM_BWD, mfn_mem_bwd_launch(&md, &H, s));
  }
  MfnAttFused F;
  if (mfn_fused_desc(P, params, W, F)) {
    // one launch: dh2, d attended, softmax backward, dh1, d cStar and its scatter onto the LSTMs' dc (added into the zero block)
    RUN(K_MFN_ATT_BWD, mfn_att_fused_bwd_launch(F, s));
  } else {
    // fp32 plans with few rows: the four input-gradient products as row-block launches too (lin_rows.hip, trans = 1)
    long lr_max = 5120;
    if (const char* e = opt_get("MFM_LIN_ROWS_MAXROWS")) lr_max = atol(e);
    const bool lr_on = prec == 0 && TB <= lr_max && !(opt_get("MFM_LIN_ROWS") && atoi(opt_get("MFM_LIN_ROWS")) == 0);
    auto rows = [&](const MfmGemmDesc& d, int kind, float* aux) {
      LinRowsItem it;
      memset(&it, 0, sizeof(it));
      it.a = d.a; it.lda = (int)d.a_sm; it.w = d.b; it.ldw = (int)d.b_sk; it.c = d.c; it.ldc = (int)d.ldc;
      it.n = d.n; it.k = d.k; it.kind = kind; it.aux = aux; it.trans = 1; it.accumulate = d.accumulate;
      return it;
    };
    {   // dh2 = d(pre cHat) W_att2_fc2, times the relu / dropout mask of att2_fc1's output
      MfmGemmDesc g = nn(W + P->dchat, M, M, PW(P, params, pi.att2_2), c.nn2, c.nn2, W + P->dh2, c.nn2, TB, 0);
      GemmEpi e = {W + P->m2, 0.0f, 3, 0u, 0};
      LinRowsItem it = rows(g, 3, e.aux);
      es.epi = &e; es.count = 1;
      if (lr_on && lin_rows_supported(&it, 1, (int)TB)) RUN(K_MFN_ATT_BWD, lin_rows_launch(&it, 1, (int)TB, 1, 0ull, s));
      else RUN(K_MFN_ATT_BWD, gemm_group_launch(&g, 1, s, nullptr, nullptr, 0, prec, &es));
    }
    {   // d attended = dh2 W_att2_fc1 + du1 W_gamma1_fc1[:, :A2] + du2 W_gamma2_fc1[:, :A2]   (into the zero-filled buffer)
      MfmGemmDesc g[3];
      g[0] = nn(W + P->dh2, c.nn2, c.nn2, PW(P, params, pi.att2_1), A2, A2, W + P->datt, A2, TB, 1);
      g[1] = nn(W + P->du1, c.g1, c.g1, PW(P, params, pi.g1_1), A2 + M, A2, W + P->datt, A2, TB, 1);
      g[2] = nn(W + P->du2, c.g2, c.g2, PW(P, params, pi.g2_1), A2 + M, A2, W + P->datt, A2, TB, 1);
      LinRowsItem it[3] = {rows(g[0], 0, nullptr), rows(g[1], 0, nullptr), rows(g[2], 0, nullptr)};
      if (lr_on && lin_rows_supported(it, 3, (int)TB)) RUN(K_MFN_ATT_BWD, lin_rows_launch(it, 3, (int)TB, 1, 0ull, s));
      else RUN(K_MFN_ATT_BWD, gemm_group_launch(g, 3, s, nullptr, nullptr, 0, prec));
    }
    RUN(K_MFN_GLUE, mfn_softmax_bwd_launch(W + P->datt, W + P->att, W + P->cstar, W + P->dlog, W + P->dcs, TB, A2, s));
    {   // dh1 = d logits W_att1_fc2, times the mask of att1_fc1's output
      MfmGemmDesc g = nn(W + P->dlog, A2, A2, PW(P, params, pi.att1_2), c.nn1, c.nn1, W + P->dh1, c.nn1, TB, 0);
      GemmEpi e = {W + P->m1, 0.0f, 3, 0u, 0};
      LinRowsItem it = rows(g, 3, e.aux);
      es.epi = &e; es.count = 1;
      if (lr_on && lin_rows_supported(&it, 1, (int)TB)) RUN(K_MFN_ATT_BWD, lin_rows_launch(&it, 1, (int)TB, 1, 0ull, s));
      else RUN(K_MFN_ATT_BWD, gemm_group_launch(&g, 1, s, nullptr, nullptr, 0, prec, &es));
    }
    {   // d cStar += dh1 W_att1_fc1   (on top of the softmax kernel's d attended * attention)
      MfmGemmDesc g = nn(W + P->dh1, c.nn1, c.nn1, PW(P, params, pi.att1_1), A2, A2, W + P->dcs, A2, TB, 1);
      LinRowsItem it = rows(g, 0, nullptr);
      if (lr_on && lin_rows_supported(&it, 1, (int)TB)) RUN(K_MFN_ATT_BWD, lin_rows_launch(&it, 1, (int)TB, 1, 0ull, s));
      else RUN(K_MFN_ATT_BWD, gemm_group_launch(&g, 1, s, nullptr, nullptr, 0, prec));
    }
    {   // d cStar -> d c_t of the three LSTMs
      MfnCs cs;
      memset(&cs, 0, sizeof(cs));
      for (int m = 0; m < 3; ++m) { cs.dcx[m] = W + P->dcx[m]; cs.h[m] = P->enc[3 + m].h; }
      cs.T = T; cs.B = B;
      RUN(K_MFN_GLUE, mfn_dcs_scatter_launch(cs, W + P->dcs, s));
    }
  }
  // ---- weight gradients of the MFN Linears (sums over all T*B rows; biases = column sums)
  float* G = grads;
  const int64_t* o = P->off;
  tn(W + P->dh1, c.nn1, c.nn1, W + P->cstar, A2, A2, G + o[pi.att1_1], A2, TB);       colsum(W + P->dh1, c.nn1, c.nn1, G + o[pi.att1_1 + 1], TB);
  tn(W + P->dlog, A2, A2, W + P->h1, c.nn1, c.nn1, G + o[pi.att1_2], c.nn1, TB);      colsum(W + P->dlog, A2, A2, G + o[pi.att1_2 + 1], TB);
  tn(W + P->dh2, c.nn2, c.nn2, W + P->attended, A2, A2, G + o[pi.att2_1], A2, TB);    colsum(W + P->dh2, c.nn2, c.nn2, G + o[pi.att2_1 + 1], TB);
  tn(W + P->dchat, M, M, W + P->h2, c.nn2, c.nn2, G + o[pi.att2_2], c.nn2, TB);       colsum(W + P->dchat, M, M, G + o[pi.att2_2 + 1], TB);
  const int64_t du[2] = {P->du1, P->du2}, dzb[2] = {P->gam1, P->gam2}, ab[2] = {P->a1, P->a2};
  const int gw[2] = {c.g1, c.g2}, gi1[2] = {pi.g1_1, pi.g2_1}, gi2[2] = {pi.g1_2, pi.g2_2};
  for (int n = 0; n < 2; ++n) {
    // gamma_n_fc1 = [attended columns | memory columns]: the memory part multiplies mem_{t-1} (zero at t = 0)
    tn(W + du[n], gw[n], gw[n], W + P->attended, A2, A2, G + o[gi1[n]], A2 + M, TB);
    if (T > 1) tn(W + du[n] + (int64_t)B * gw[n], gw[n], gw[n], W + P->mems, M, M, G + o[gi1[n]] + A2, A2 + M, TB - B);
    colsum(W + du[n], gw[n], gw[n], G + o[gi1[n] + 1], TB);
    tn(W + dzb[n], M, M, W + ab[n], gw[n], gw[n], G + o[gi2[n]], gw[n], TB);           colsum(W + dzb[n], M, M, G + o[gi2[n] + 1], TB);
  }
  return MFM_OK;
}

// Descriptor table and block list of the weight-gradient role workgroups (dw_role_dev.h).  The block list depends on the
// products' shapes and on which buffer their A operand lives in, not on addresses that change per call: it is built once per
// (plan, stage form) and uploaded into the workspace.  MFM_ERR_UNSUPPORTED: a product the role blocks do not take.
static int dw_role_build(MfmPlan* P, const std::vector<MfmGemmDesc>& all, float* W, int key, hipStream_t s, DwRole* out) {
  const int T = P->T, B = P->B;
  const int64_t TB = (int64_t)T * B;
  const int n = (int)all.size();
  if (n < 1 || n > DWR_MAXP || !gemm_tn_supported(all.data(), n, INT32_MAX, true)) return MFM_ERR_UNSUPPORTED;
  DwRole& DR = *out;
  memset(&DR, 0, sizeof(DR));
  DR.count = n;
  int dep[DWR_MAXP], tbase[DWR_MAXP];
  const int64_t lim = (int64_t)1 << 29;
  for (int i = 0; i < n; ++i) {
    const MfmGemmDesc& d = all[i];
    if (d.a_bf16 || d.c_bf16 || gemm_get_colsum_host(d)) return MFM_ERR_UNSUPPORTED;
    if (d.a_sz >= lim || d.b_sz >= lim || d.c_sz >= lim || d.a_sk >= lim || d.b_sk >= lim || d.ldc >= lim) return MFM_ERR_UNSUPPORTED;
    DwRoleProblem& q = DR.p[i];
    q.a = d.a; q.b = d.b; q.c = d.c; q.c2 = d.c2;
    q.a_sz = (int)d.a_sz; q.b_sz = (int)d.b_sz; q.c_sz = (int)d.c_sz; q.a_sk = (int)d.a_sk; q.b_sk = (int)d.b_sk; q.ldc = (int)d.ldc;
    q.m = d.m; q.n_valid = (d.n_valid <= 0 || d.n_valid > d.n) ? d.n : d.n_valid; q.k = d.k; q.batch = d.batch; q.alpha = d.alpha;
    q.b_shift = 0;
    dep[i] = DWR_DEP_NONE; tbase[i] = 0;
    // products over an LSTM's gate gradients: which buffer (the encoders' are written inside the launch) and, for the
    // recurrent product sum_{t >= 1} dA_t^T h_{t-1} -- A one time step into the buffer -- the SAME rows as the input and
    // bias products of that LSTM with B shifted instead, so that the three share their A slices (dw_role_dev.h)
    for (int e = 0; e < 7; ++e) {
      const SeqBuf& sb = e < 4 ? P->enc[e] : P->dec[e - 4];
      const float* g0 = W + sb.gates;
      const int64_t step = (int64_t)B * 4 * sb.Hp;
      if (d.a >= g0 && d.a < g0 + TB * 4 * sb.Hp) {
        const int tb = (int)((d.a - g0) / step);
        if (e < 4) { dep[i] = e + 1; tbase[i] = tb; }
        if (tb == 1 && d.a == g0 + step && d.k == (int)(TB - B)) {
          q.a = g0; q.k = (int)TB; q.b_shift = B;
          if (e < 4) tbase[i] = 0;
        }
      }
    }
    q.tiles_m = cdiv(d.m, DWR_T); q.tiles_n = cdiv(d.n, DWR_T);
    const int split = cdiv(q.k, DWR_KC);
    q.kps = round_up(cdiv(q.k, split), 4);
    if (d.a >= W + P->lat_grd && d.a < W + P->lat_grd + (int64_t)B * P->lat.rec_size) dep[i] = DWR_DEP_LATENT;
    if (opt_get("MFM_DW_FOLD_NODEP")) dep[i] = DWR_DEP_NONE;       // timing experiment only (wrong gradients): nothing waits
  }
  int n_role = device_cus() - 4 * B;
  if (const char* e = opt_get("MFM_DW_FOLD_ROLES")) { const int v = atoi(e); if (v >= 1 && v <= n_role) n_role = v; }
  if (n_role < 1) return MFM_ERR_UNSUPPORTED;
  DR.n_role = n_role;
  const int nslots = 4 * n_role;
  if (P->dw_table_key != key || P->dw_table_host.empty()) {
    // tiles that read the same A slice -- same operand, gate block z, row tile tm, and therefore the same chunks -- form a
    // GROUP; a role workgroup takes up to four tiles of one group per iteration (its four slots share the A image)
    struct Group { const float* a; int a_sk, a_sz, k, kps, m, z, tm, dep, tbase; std::vector<std::pair<int, int>> tiles; };   // tiles: (problem, tile id)
    std::vector<Group> groups;
    for (int i = 0; i < n; ++i) {
      const DwRoleProblem& q = DR.p[i];
      for (int z = 0; z < q.batch; ++z)
        for (int tm = 0; tm < q.tiles_m; ++tm) {
          Group* g = nullptr;
          for (auto& c : groups)
            if (c.a == q.a && c.a_sk == q.a_sk && c.a_sz == q.a_sz && c.k == q.k && c.kps == q.kps && c.m == q.m && c.z == z &&
                c.tm == tm && c.dep == dep[i] && c.tbase == tbase[i]) { g = &c; break; }
          if (!g) { groups.push_back(Group{q.a, q.a_sk, q.a_sz, q.k, q.kps, q.m, z, tm, dep[i], tbase[i], {}}); g = &groups.back(); }
          for (int tn = 0; tn < q.tiles_n; ++tn) g->tiles.push_back({i, tn + q.tiles_n * (tm + q.tiles_m * z)});
        }
    }
    // a workgroup item: up to four tiles of one group
    struct Item { int grp, first, count, chunk; };
    // phase A: one item per (tile set, chunk) of the groups whose A operand does not come from the encoder BPTT, partial
    // tiles added with atomics: operands that are final before the launch first, the latent stack's behind them
    std::vector<Item> ua;
    for (int pass = 0; pass < 2; ++pass)
      for (size_t gi = 0; gi < groups.size(); ++gi) {
        const Group& g = groups[gi];
        if (g.dep != (pass == 0 ? DWR_DEP_NONE : DWR_DEP_LATENT)) continue;
        const int split = cdiv(g.k, g.kps);
        for (int sp = 0; sp < split; ++sp)
          for (int f = 0; f < (int)g.tiles.size(); f += 4) ua.push_back({(int)gi, f, std::min(4, (int)g.tiles.size() - f), sp});
      }
    const int rows_a = cdiv((int)ua.size(), n_role);
    // phase B: every encoder tile set stays with one workgroup for all its chunks (last time steps first); each tile is
    // accumulated in registers and written once -- a plain store, the gradient buffer holds zeros and nobody else adds there
    std::vector<Item> te;
    int max_split = 0;
    for (size_t gi = 0; gi < groups.size(); ++gi) {
      const Group& g = groups[gi];
      if (g.dep < 1 || g.dep > 4) continue;
      max_split = std::max(max_split, cdiv(g.k, g.kps));
      for (int f = 0; f < (int)g.tiles.size(); f += 4) te.push_back({(int)gi, f, std::min(4, (int)g.tiles.size() - f), 0});
    }
    const int nacc = cdiv((int)te.size(), n_role);
    if (nacc > 2) return MFM_ERR_UNSUPPORTED;
    const int n_iter = rows_a + nacc * max_split;
    if ((int64_t)n_iter * nslots > DWR_TABLE_CAP) return MFM_ERR_UNSUPPORTED;
    // B > 32 (round 4: stamps for up to 64 rows): fewer idle CUs carry more work.  A role workgroup gets through one block per
    // ~5.5 us (a chain of memory round trips) and the BPTT it hides behind lasts ~1.45 us per time step whatever B is: beyond
    // ~5 + 0.3 T blocks per workgroup the launch ends later than BPTT + separate launch would.  Measured, ms per step, role form
    // vs separate launch: MOSI T = 20: B = 33 0.173 / 0.182, 36 0.177 / 0.188, 38 0.181 / 0.189 (11 blocks), 39 0.219 / 0.190,
    // 40 0.218 / 0.190, 48 0.256 / 0.197 (a block is bound by the ~80 KB it pulls through the CU, not by latency: requesting the
    // next block's operands during the product gained 1-2 %); YouTube shape B = 36: T = 35 0.308 / 0.258, T = 10 0.149 / 0.144 (two
    // accumulator rounds: excluded by the rule).  B <= 32 always takes the role form (MFM_DW_FOLD_MAXITER overrides)
    if (B > 32 || opt_get("MFM_DW_FOLD_MAXITER")) {
      int max_iter = 5 + (3 * T) / 10;
      if (const char* e = opt_get("MFM_DW_FOLD_MAXITER")) max_iter = atoi(e);
      if (n_iter > max_iter) return MFM_ERR_UNSUPPORTED;
    }
    P->dw_table_host.assign((size_t)n_iter * nslots * 4, 0);
    for (size_t i = 0; i < (size_t)n_iter * nslots; ++i) P->dw_table_host[4 * i] = -1;
    auto put = [&](int row, int wg, const Item& it, int chunk, int w) {
      const Group& g = groups[it.grp];
      for (int s4 = 0; s4 < it.count; ++s4) {
        int* e = &P->dw_table_host[((size_t)row * nslots + 4 * wg + s4) * 4];
        e[0] = g.tiles[it.first + s4].first; e[1] = g.tiles[it.first + s4].second; e[2] = chunk; e[3] = w;
      }
    };
    for (size_t u = 0; u < ua.size(); ++u)
      put((int)(u / n_role), (int)(u % n_role), ua[u], ua[u].chunk, groups[ua[u].grp].dep | DWR_FIRST | DWR_LAST);
    const bool store_ok = !opt_get("MFM_DW_FOLD_ATOMICS");           // (A/B timing: MFM_DW_FOLD_ATOMICS=1 keeps the atomics)
    for (size_t j = 0; j < te.size(); ++j) {
      const Group& g = groups[te[j].grp];
      const int split = cdiv(g.k, g.kps);
      const int wg = (int)(j % n_role), acc = (int)(j / n_role);
      for (int c = 0; c < split; ++c) {            // c-th block of this tile set: chunk split - 1 - c
        const int sp = split - 1 - c;
        const int t0 = g.tbase + (sp * g.kps) / B;
        int w = g.dep | (t0 << 8) | (acc ? DWR_ACC1 : 0) | (store_ok ? DWR_STORE : 0);
        if (c == 0) w |= DWR_FIRST;
        if (c == split - 1) w |= DWR_LAST;
        put(rows_a + (max_split - split + c) * nacc + acc, wg, te[j], sp, w);
      }
    }
    P->dw_table_key = key;
    P->dw_table_ws = nullptr;
  }
  if (P->dw_table_ws != W) {
    MFM_REQUIRE(!stream_capturing(s), "plan: the first backward of a plan uploads its weight-gradient block table from host memory, which "
                                      "cannot be captured into a hipGraph -- run one eager step on this plan before capturing");
    MFM_HIP_CHECK(hipMemcpyAsync(W + P->dw_table, P->dw_table_host.data(), P->dw_table_host.size() * sizeof(int), hipMemcpyHostToDevice, s));
    P->dw_table_ws = W;
  }
  DR.n_iter = (int)(P->dw_table_host.size() / 4 / nslots);
  DR.any_dep = 0;
  for (int i = 0; i < n; ++i) DR.any_dep |= (dep[i] != DWR_DEP_NONE);
  DR.table = reinterpret_cast<const int4*>(W + P->dw_table);
  return MFM_OK;
}

static int backward(MfmPlan* P, const float* params, const float* x, const void* y, int stage, float* W,
                    float* grads, hipStream_t s, const ExtGrads* ext = nullptr, const LossW* lw = nullptr) {
  OptScope _opts(P->opts);
  const MfmPlanConfig& c = P->cfg;
  const int V = c.variant;
  const int T = P->T, B = P->B;
  const int64_t TB = (int64_t)T * B;
  if (P->grads_prezeroed != grads) MFM_HIP_CHECK(hipMemsetAsync(grads, 0, (size_t)P->n_params * sizeof(float), s));
  P->grads_prezeroed = nullptr;
  // the guard word of this gradient buffer (plan option "grad_guard_offset"): NaN while the plan's status word is set
  float* const guard = (P->opt_guard >= 0 && P->opt_guard < P->n_params) ? grads + P->opt_guard : nullptr;
  struct GuardAtExit {      // the role-workgroup launch does it itself; every other way out of this function: one tiny launch,
    MfmPlan* P; float* W; float* guard; hipStream_t s; bool armed;      // only on plans that ever used a hand-over
    ~GuardAtExit() {
      if (armed && guard && P->ever_handover)
        hipLaunchKernelGGL(guard_propagate_kernel, dim3(1), dim3(64), 0, s, P->status_ptr(W), guard);
    }
  } guard_at_exit{P, W, guard, s, true};
  const bool gen_on = lw ? lw->gen_on != 0 : (stage != 2), disc_on = lw ? lw->disc != 0.0f : (stage != 1);
  const bool seq_bf16 = P->seq_bf16;
  const bool st16 = P->st16;
  MFM_REQUIRE(!(ext && st16), "plan: backward for external upstream gradients is not available on a bf16-resident plan "
                              "(the module path runs fp32 plans)");
  // bf16-resident plans: every sum over the T*B rows that feeds an LSTM's or a decoder fc1's weight gradient is an item of
  // ONE dw_bf16_kernel launch behind the encoder BPTT
  DwbLaunch DB;
  memset(&DB, 0, sizeof(DB));
  DB.rows = (int)TB;
  // every weight-gradient product only feeds the optimizer: they are collected here and issued as ONE grouped
  // launch behind the encoder BPTT (49 problems at the canonical wiring) instead of three launches on the chain
  std::vector<MfmGemmDesc> tail;
  if (gen_on) {
    // B0: through decoder fc1
    std::vector<MfmGemmDesc> g;
    for (int m = 0; m < 3; ++m) {
      const SeqBuf& sb = P->dec[m];
      const int pb = P->dec_p[m];
      MfmGemmDesc d;
      memset(&d, 0, sizeof(d));
      d.alpha = 1.0f; d.batch = 1;
      // dH = dx_hat Wfc  (pad units -> exact zeros)
      const float* dxh = (ext && ext->d_xhat[m]) ? ext->d_xhat[m] : W + P->dxhat[m];
      d.a = dxh; d.a_sm = st16 ? P->dxh_ld[m] : P->dec_d[m]; d.a_sk = 1; d.a_bf16 = st16 ? 1 : 0;
      d.b = params + P->off[pb + FC_W]; d.b_sk = sb.h; d.b_sn = 1;
      d.c = W + P->dec_dhs[m]; d.ldc = sb.Hp; d.c_bf16 = st16 ? 1 : 0;
      d.m = (int)TB; d.n = sb.Hp; d.n_valid = sb.h; d.k = P->dec_d[m]; d.split_k = 1;
      g.push_back(d);
      if (st16) {
        // dWfc = dx_hat^T H and dbfc = column sums of dx_hat: one item of the one-pass launch
        DwbItem& I = DB.it[DB.n_items++];
        I.a = reinterpret_cast<const __bf16*>(W + P->dxhat[m]); I.lda = P->dxh_ld[m]; I.M = P->dec_d[m];
        I.Hp = P->dxh_ld[m]; I.h = P->dec_d[m];
        I.nseg = 1; I.seg[0].p = reinterpret_cast<const __bf16*>(W + sb.hs); I.seg[0].ld = sb.Hp; I.seg[0].ncols = sb.Hp;
        I.seg[0].col0 = 0; I.seg[0].shift = 0; I.seg[0].rows = (int)TB;
        I.nout = 1; I.out[0].n0 = 0; I.out[0].nvalid = sb.h; I.out[0].c = grads + P->off[pb + FC_W]; I.out[0].ldc = sb.h;
        I.cb = grads + P->off[pb + FC_B];
        continue;
      }
      // dWfc = dx_hat^T H
      MfmGemmDesc w;
      memset(&w, 0, sizeof(w));
      w.alpha = 1.0f; w.batch = 1; w.accumulate = 1; w.split_k = 0;
      w.a = dxh; w.a_sm = 1; w.a_sk = P->dec_d[m];
      w.b = W + sb.hs; w.b_sk = sb.Hp; w.b_sn = 1;
      w.c = grads + P->off[pb + FC_W]; w.ldc = sb.h;
      w.m = P->dec_d[m]; w.n = sb.h; w.n_valid = sb.h; w.k = (int)TB;
      tail.push_back(w);
      // dbfc = column sums of dx_hat
      MfmGemmDesc bb = w;
      bb.b = W + P->ones; bb.b_sk = 1; bb.b_sn = 1;
      bb.c = grads + P->off[pb + FC_B]; bb.ldc = 1; bb.n = 1; bb.n_valid = 1;
      tail.push_back(bb);
    }
    // dH is already there when this step's forward ran the fused fc1 kernel and the gradient is the plan's own d x_hat
    const bool dh_done = !ext && P->fc1_bwd_call == P->calls;
    if (!dh_done) RUN(K_FC1_BWD, gemm_group_launch(g.data(), (int)g.size(), s, nullptr, nullptr, 0, c.precision));
    // B1: decoder BPTT
    {
      MfmSeqDesc q[3];
      for (int m = 0; m < 3; ++m) {
        q[m] = seq_desc(P, P->dec[m], P->dec_p[m], params, W, true);
        q[m].h_init = W + P->dec_init[m]; q[m].ld_init = P->dec_h[m];
        q[m].dh_ext = W + P->dec_dhs[m]; q[m].ld_dh = P->dec[m].Hp;
        q[m].d_h_init = W + P->dec_dinit[m]; q[m].ld_dinit = P->dec_h[m];
      }
      const bool imgs_on = !seq_bf16 && P->wt_call == P->calls;
      const int ne = P->n_enc;
      const float* dimg[3] = {imgs_on ? W + P->wt_img[ne] : nullptr, imgs_on ? W + P->wt_img[ne + 1] : nullptr, imgs_on ? W + P->wt_img[ne + 2] : nullptr};
      if (imgs_on) RUN(K_DEC_BWD, seq_bwd_img_launch(q, 3, T, B, dimg, s));
      else RUN(K_DEC_BWD, seq_bf16 ? mfm_lstm_seq_bwd_bf16(q, 3, T, B, s) : mfm_lstm_seq_bwd(q, 3, T, B, s));
    }
    // (B2: the decoder weight gradients only feed Adam; they share the encoders' launch at the end)
  }
  // B3: latent stack
  bool enc_bwd_done = false;
  {
    LatentDev L = P->lat;
    L.ops = reinterpret_cast<const LatOp*>(W + P->lat_ops_off);
    L.items_fwd = reinterpret_cast<const int*>(W + P->lat_items_off);
    L.items_bwd = L.items_fwd + (size_t)4 * MFM_LAT_MAXSTAGES * MFM_LAT_ROW_THREADS * 4;
    if (opt_get("MFM_LATENT_DBG")) L.dbg = reinterpret_cast<unsigned long long*>(W + P->dbg_off);
    for (int m = 0; m < 3; ++m) {
      L.d_dec_init[m] = gen_on ? W + P->dec_dinit[m] : nullptr;
      L.dec_ld[m] = P->dec_h[m];
    }
    for (int e = 0; e < 4; ++e) {
      L.dh_last[e] = W + P->dh_last[e];
      L.dh_ld[e] = (e == 3 && V != 0) ? P->nzy : P->enc_h[e];
    }
    L.rec = W + P->lat_rec;
    L.y = y;
    L.grd_out = W + P->lat_grd;
    if (V == 2) {        // d MMD / d z, written by the forward; its weight: lda_mmd, or the caller's upstream gradient
      L.grd_seed = W + P->lat_seed;
      L.seed_w = lw ? lw->reg : c.lda_reg; L.seed_w_ptr = ext ? ext->d_reg : nullptr;
    }
    if (ext) { L.d_yhat_ext = ext->d_yhat; L.reg_w_ptr = ext->d_reg; }
    L.reg_w = (lw ? lw->reg : c.lda_reg) * c.reg_scale;
    L.disc_w = lw ? lw->disc : (disc_on ? 1.0f : 0.0f);
    L.disc_loss_out = (lw && lw->write_disc && y) ? W + P->losses : nullptr;
    L.gen_w = gen_on ? 1.0f : 0.0f;
    // weight gradients of the 22 latent Linears: dW[n][k] = sum_r G[r][out+n] X[r][in+k]
    const int rs = P->lat.rec_size;
    auto latent_products = [&](std::vector<MfmGemmDesc>& out, bool colsum) {
      for (int i = 0; i < P->lat.nops; ++i) {
        const LatOp& op = P->lat_ops[i];
        MfmGemmDesc d;
        memset(&d, 0, sizeof(d));
        d.alpha = 1.0f; d.batch = 1; d.split_k = 1; d.accumulate = 0;
        d.a = W + P->lat_grd + op.out_off; d.a_sm = 1; d.a_sk = rs;
        d.b = W + P->lat_rec + op.in_off; d.b_sk = rs; d.b_sn = 1;
        d.c = grads + op.w_off; d.ldc = op.K;
        d.m = op.N; d.n = op.K; d.n_valid = op.K; d.k = P->B;
        if (colsum) gemm_set_colsum(d, grads + op.b_off);
        out.push_back(d);
      }
    };
    // MFM_KL_EF at small batches: the encoder BPTT workgroups run their rows' chains first (fold launch); B4 is then done too
    if (V == 0 && !seq_bf16 && P->n_enc == 4 && P->fold_state == 1) {
      MfmSeqDesc q[4];
      for (int e = 0; e < 4; ++e) {
        q[e] = seq_desc(P, P->enc[e], P->enc_p[e], params, W, false);
        q[e].dh_ext = W + P->dh_last[e]; q[e].ld_dh = P->enc_h[e];
      }
      int rc = MFM_ERR_UNSUPPORTED;
      const bool imgs_on = P->wt_call == P->calls;
      const float* eimg[4] = {imgs_on ? W + P->wt_img[0] : nullptr, imgs_on ? W + P->wt_img[1] : nullptr,
                              imgs_on ? W + P->wt_img[2] : nullptr, imgs_on ? W + P->wt_img[3] : nullptr};
      // B <= 32: the idle CUs of this launch run every weight-gradient product of the step (dw_role_dev.h); B5 disappears
      if (!st16 && P->dwfold_state >= 0 && P->dw_table >= 0 && P->opt_handover && seq_small_folddw_supported(T, B) &&
          !opt_get("MFM_DW_ONEPASS_MINROWS") && !opt_get("MFM_DW_F32_MINROWS") && !(opt_get("MFM_GEMM_TN") && atoi(opt_get("MFM_GEMM_TN")) == 0)) {
        std::vector<MfmGemmDesc> all = tail;
        latent_products(all, false);
        for (int e = 0; e < 4; ++e) dA_gemms(P, P->enc[e], P->enc_p[e], W, grads, all, x + P->enc_xoff[e], P->D, P->enc_d[e], false);
        if (gen_on)
          for (int m = 0; m < 3; ++m) dA_gemms(P, P->dec[m], P->dec_p[m], W, grads, all, W + P->dec_init[m], P->dec_h[m], P->dec_h[m], true, false);
        DwRole DR;
        const int key = (gen_on ? 1 : 0) | ((disc_on || L.disc_loss_out) ? 2 : 0) | (ext ? 4 : 0);
        const int brc = dw_role_build(P, all, W, key, s, &DR);
        if (brc == MFM_OK) {
          LatentDev L2 = L;
          L2.grd_agent = 1;
          DR.flags = reinterpret_cast<unsigned*>(W + P->dw_flags); DR.epoch = epoch_base(++P->dw_epoch); DR.tick = P->dw_tick_ptr(W);
          DR.bf16 = c.precision ? 1 : 0;
          // a block that gives up: status bit 1, NaN into the gradient guard (no guard word: into the first gradient)
          DR.ctl = P->ho_ctl(W, guard ? guard : grads, 2u);
          DR.fault = (P->opt_fault == 2) ? 1 : 0;
          if (DR.fault) P->opt_fault = 0;
          { Timer _t(P, s, K_ENC_BWD); rc = seq_folddw_launch(q, 4, T, B, L2, params, grads, DR, s, imgs_on ? eimg : nullptr); }
          if (rc == MFM_OK) {             // every gradient of the step is on its way
            P->dwfold_state = 1; P->ever_handover = true;
            guard_at_exit.armed = false;
            if (stream_capturing(s)) {
              hipLaunchKernelGGL(tick_kernel, dim3(1), dim3(64), 0, s, (unsigned long long*)nullptr, P->dw_tick_ptr(W));
              MFM_LAUNCH_CHECK("tick_kernel");
            }
            return MFM_OK;
          }
          if (rc != MFM_ERR_UNSUPPORTED) return rc;
        } else if (brc != MFM_ERR_UNSUPPORTED) return brc;
        P->dwfold_state = -1;
      }
      { Timer _t(P, s, K_ENC_BWD); rc = seq_fold_launch(q, 4, T, B, true, L, params, grads, s, imgs_on ? eimg : nullptr); }
      if (rc == MFM_OK) enc_bwd_done = true;
      else if (rc != MFM_ERR_UNSUPPORTED) return rc;
    }
    // bf16 plans from B = 192 send the latent weight gradients to gemm_tn_kernel (end of this function): the staged backward
    // then leaves the bias gradients to that launch's column sums instead of adding 1180 words per workgroup into the same
    // addresses (10 of its 60 us at B = 2048, profiles/r03_latent_mfma.txt); MFM_LATENT_BIAS_TN=0 keeps the atomics
    bool bias_in_tail = false;
    if (!enc_bwd_done && !L.row_path && c.precision && B <= 8192 && !(opt_get("MFM_GEMM_TN") && atoi(opt_get("MFM_GEMM_TN")) == 0) &&
        !(opt_get("MFM_LATENT_BIAS_TN") && atoi(opt_get("MFM_LATENT_BIAS_TN")) == 0)) {
      const long minb = opt_get("MFM_GEMM_TN_BF16_MINB") ? atol(opt_get("MFM_GEMM_TN_BF16_MINB")) : 192;
      long rows16 = 8192;
      if (const char* e = opt_get("MFM_GEMM_TN_MAXROWS_BF16")) rows16 = atol(e);
      bias_in_tail = B >= minb && B <= rows16;
    }
    L.skip_bias = bias_in_tail ? 1 : 0;
    if (!enc_bwd_done) RUN(K_LAT_BWD, latent_bwd_launch(L, params, grads, s));
    latent_products(tail, bias_in_tail);
  }
  // Memory Fusion Network (variants 1, 2): from d [mu_y | logvar_y] back to d h_T / d c_t of its three LSTMs
  if (V != 0) {
    int rc = mfn_backward(P, params, W, grads, s, tail);
    if (rc != MFM_OK) return rc;
  }
  // B4: encoder BPTT (up to MFM_MAX_SEQ per launch)
  for (int e0 = 0; e0 < P->n_enc && !enc_bwd_done; e0 += MFM_MAX_SEQ) {
    MfmSeqDesc q[MFM_MAX_SEQ];
    const int n = std::min(MFM_MAX_SEQ, P->n_enc - e0);
    for (int i = 0; i < n; ++i) {
      const int e = e0 + i;
      q[i] = seq_desc(P, P->enc[e], P->enc_p[e], params, W, false);
      if (V != 0 && e >= 3) {     // MFN LSTM: gradient on h_{T-1} from the heads, on every c_t from the attention block
        int hoff = 0;
        for (int m = 0; m < e - 3; ++m) hoff += P->enc[3 + m].h;
        q[i].dh_ext = W + P->d_hT + hoff; q[i].ld_dh = P->tot;
        q[i].dc_ext = W + P->dcx[e - 3];
      } else {
        q[i].dh_ext = W + P->dh_last[e]; q[i].ld_dh = P->enc_h[e];
      }
    }
    if (!seq_bf16 && P->wt_call == P->calls && e0 == 0 && n == P->n_enc) {
      const float* eimg[MFM_MAX_SEQ];
      for (int i = 0; i < n; ++i) eimg[i] = W + P->wt_img[i];
      RUN(K_ENC_BWD, seq_bwd_img_launch(q, n, T, B, eimg, s));
    } else RUN(K_ENC_BWD, seq_bf16 ? mfm_lstm_seq_bwd_bf16(q, n, T, B, s) : mfm_lstm_seq_bwd(q, n, T, B, s));
  }
  // B5: all weight gradients on the grouped TN GEMM.  Opt-in (MFM_DW_ONEPASS_MINROWS=<T*B from which to use it>): the
  // LSTMs' sums over the rows on the one-pass kernel (dw_onepass.hip) -- parity-tested, measured slower at B=2048
  {
    long dw_min_rows = 1L << 60;
    if (const char* e = opt_get("MFM_DW_ONEPASS_MINROWS")) dw_min_rows = atol(e);
    const bool onepass = !st16 && TB >= dw_min_rows && (int64_t)TB * P->D < ((int64_t)1 << 29);
    if (st16) {
      // the batch as bf16, modality slices on 16-column boundaries (what the one-pass kernel streams by LDS-DMA)
      const int src0[3] = {0, c.d_l, c.d_l + c.d_a}, nn[3] = {c.d_l, c.d_a, c.d_v};
      if (P->x16_call != P->calls) RUN(K_PACK, x_to_bf16_launch(x, W + P->x16, TB, P->D, P->x16_ld, src0, nn, P->x16_off, s));
      auto lstm_item = [&](const SeqBuf& sb, int pb, int xcol0, int xcols, bool dec, int e) {
        DwbItem& I = DB.it[DB.n_items++];
        I.a = reinterpret_cast<const __bf16*>(W + sb.gates); I.lda = 4 * sb.Hp; I.M = 4 * sb.Hp; I.Hp = sb.Hp; I.h = sb.h;
        int n = 0;
        if (!dec) {
          DwbSeg& S = I.seg[I.nseg++];
          S.p = reinterpret_cast<const __bf16*>(W + P->x16); S.ld = P->x16_ld; S.col0 = xcol0; S.ncols = xcols; S.shift = 0; S.rows = (int)TB;
          // output columns: one range per modality slice inside [xcol0, xcol0 + xcols)
          const int dd_[3] = {c.d_l, c.d_a, c.d_v};
          int dst = 0;
          for (int m = 0; m < 3; ++m) {
            if (P->x16_off[m] < xcol0 || P->x16_off[m] >= xcol0 + xcols) continue;
            DwbOut& O = I.out[I.nout++];
            O.n0 = P->x16_off[m] - xcol0; O.nvalid = dd_[m]; O.c = grads + P->off[pb + W_IH] + dst; O.ldc = P->enc_d[e];
            dst += dd_[m];
          }
          n = xcols;
        }
        DwbSeg& S = I.seg[I.nseg++];
        S.p = reinterpret_cast<const __bf16*>(W + sb.hs); S.ld = sb.Hp; S.col0 = 0; S.ncols = sb.Hp; S.shift = B; S.rows = (int)TB;
        DwbOut& O = I.out[I.nout++];
        O.n0 = n; O.nvalid = sb.h; O.c = grads + P->off[pb + W_HH]; O.ldc = sb.h;
        if (dec) O.c2 = grads + P->off[pb + W_IH];          // steps >= 1 feed h back as the input (mfm_model.py:85)
        I.cb = grads + P->off[pb + B_IH]; I.cb2 = grads + P->off[pb + B_HH];
      };
      for (int e = 0; e < P->n_enc; ++e) {
        const bool whole = (V == 0 && e == 3);                 // the early-fusion encoder consumes every slice
        const int mod = whole ? 0 : (e < 3 ? e : e - 3);
        lstm_item(P->enc[e], P->enc_p[e], whole ? 0 : P->x16_off[mod], whole ? P->x16_ld : round_up(P->enc_d[e], 16), false, e);
      }
      if (gen_on)
        for (int m = 0; m < 3; ++m) {
          lstm_item(P->dec[m], P->dec_p[m], 0, 0, true, 0);
          dA_gemms(P, P->dec[m], P->dec_p[m], W, grads, tail, W + P->dec_init[m], P->dec_h[m], P->dec_h[m], true, true);
        }
      MFM_REQUIRE(DB.n_items <= MFM_DWB_MAXI, "plan: %d one-pass items", DB.n_items);
      if (P->dwb_slabs >= 0) { DB.slabs = W + P->dwb_slabs; DB.slab_floats = P->dwb_slab_floats; }
      RUN(K_DEC_DW, dw_bf16_launch(DB, s));
    }
    // fp32 plans at large T*B (round 3): the LSTMs' sums over the rows on the fp32 form of the one-pass kernel
    // (dw_stream_kernel<true>: the batch and the fp32 dA / h buffers streamed by LDS-DMA in memory order, exact fp32 MFMA
    // chains); MFM_DW_F32_MINROWS moves the threshold (0 = off)
    long f32_min_rows = 0;          // measured slower than the grouped GEMM (dw_bf16.hip, launcher note): opt-in
    if (const char* e = opt_get("MFM_DW_F32_MINROWS")) f32_min_rows = atol(e);
    const bool f32pass = !c.precision && !onepass && f32_min_rows > 0 && TB >= f32_min_rows && TB > 1;
    bool f32_done[9] = {false, false, false, false, false, false, false, false, false};
    if (f32pass) {
      DwbLaunch DF;
      memset(&DF, 0, sizeof(DF));
      DF.rows = (int)TB; DF.f32 = 1;
      auto lstm_item32 = [&](const SeqBuf& sb, int pb, const float* xin, int xcol0, int kin, bool dec) -> bool {
        DwbItem I;
        memset(&I, 0, sizeof(I));
        I.a = reinterpret_cast<const __bf16*>(W + sb.gates); I.lda = 4 * sb.Hp; I.M = 4 * sb.Hp; I.Hp = sb.Hp; I.h = sb.h;
        int n = 0;
        bool last_row_apart = false;
        if (!dec) {
          DwbSeg& S = I.seg[I.nseg++];
          S.p = reinterpret_cast<const __bf16*>(xin); S.ld = P->D; S.col0 = xcol0; S.ncols = round_up(kin, 16); S.shift = 0;
          // a slab wider than what is left of the row runs into the next row -- harmless, those columns are never stored --
          // but behind the LAST row it would leave the batch buffer: that row's input product goes to the tail GEMM (K = 1)
          last_row_apart = xcol0 + S.ncols > P->D;
          S.rows = last_row_apart ? (int)TB - 1 : (int)TB;
          DwbOut& O = I.out[I.nout++];
          O.n0 = 0; O.nvalid = kin; O.c = grads + P->off[pb + W_IH]; O.ldc = kin;
          n = S.ncols;
        }
        DwbSeg& S = I.seg[I.nseg++];
        S.p = reinterpret_cast<const __bf16*>(W + sb.hs); S.ld = sb.Hp; S.col0 = 0; S.ncols = sb.Hp; S.shift = B; S.rows = (int)TB;
        DwbOut& O = I.out[I.nout++];
        O.n0 = n; O.nvalid = sb.h; O.c = grads + P->off[pb + W_HH]; O.ldc = sb.h;
        if (dec) O.c2 = grads + P->off[pb + W_IH];
        I.cb = grads + P->off[pb + B_IH]; I.cb2 = grads + P->off[pb + B_HH];
        if (DF.n_items >= MFM_DWB_MAXI || !dw_bf16_supported(I, 1)) return false;
        DF.it[DF.n_items++] = I;
        if (last_row_apart) {
          MfmGemmDesc d;
          memset(&d, 0, sizeof(d));
          d.a_sz = sb.Hp; d.a_sm = 1; d.a_sk = 4 * (int64_t)sb.Hp; d.m = sb.h; d.batch = 4; d.accumulate = 1; d.split_k = 1; d.alpha = 1.0f;
          d.a = W + sb.gates + (TB - 1) * 4 * sb.Hp;
          d.b = xin + (TB - 1) * P->D + xcol0; d.b_sk = P->D; d.b_sn = 1;
          d.k = 1; d.n = kin; d.n_valid = kin;
          d.c = grads + P->off[pb + W_IH]; d.c_sz = (int64_t)sb.h * kin; d.ldc = kin;
          tail.push_back(d);
        }
        return true;
      };
      for (int e = 0; e < P->n_enc; ++e) f32_done[e] = lstm_item32(P->enc[e], P->enc_p[e], x, P->enc_xoff[e], P->enc_d[e], false);
      if (gen_on)
        for (int m = 0; m < 3; ++m) f32_done[6 + m] = lstm_item32(P->dec[m], P->dec_p[m], nullptr, 0, 0, true);
      if (DF.n_items > 0) RUN(K_DEC_DW, dw_bf16_launch(DF, s));
    }
    DwLaunch DL;
    memset(&DL, 0, sizeof(DL));
    DL.rows = (int)TB;
    auto item = [&](const SeqBuf& sb, int pb, const float* xin, int64_t ldx, int kin, bool dec) {
      DwItem I;
      memset(&I, 0, sizeof(I));
      I.dA = W + sb.gates; I.ldA = 4 * sb.Hp; I.M = 4 * sb.Hp; I.Hp = sb.Hp; I.h = sb.h;
      if (!dec) { I.x = xin; I.ldx = ldx; I.dx = kin; I.c_x = grads + P->off[pb + W_IH]; I.ldc_x = kin; }
      I.hs = W + sb.hs; I.ldh = sb.Hp; I.hN = sb.h; I.shift = B;
      I.c_h = grads + P->off[pb + W_HH]; I.ldc_h = sb.h;
      if (dec) I.c_h2 = grads + P->off[pb + W_IH];          // steps >= 1 feed h back as the input (mfm_model.py:85)
      I.c_b = grads + P->off[pb + B_IH]; I.c_b2 = grads + P->off[pb + B_HH];
      return I;
    };
    for (int e = 0; e < P->n_enc && !st16; ++e) {
      if (f32_done[e]) continue;
      DwItem I = item(P->enc[e], P->enc_p[e], x + P->enc_xoff[e], P->D, P->enc_d[e], false);
      if (onepass && DL.n_items < MFM_DW_MAXI && dw_onepass_supported(I, c.precision)) DL.it[DL.n_items++] = I;
      else dA_gemms(P, P->enc[e], P->enc_p[e], W, grads, tail, x + P->enc_xoff[e], P->D, P->enc_d[e], false);
    }
    if (gen_on && !st16)
      for (int m = 0; m < 3; ++m) {
        DwItem I = item(P->dec[m], P->dec_p[m], nullptr, 0, 0, true);
        const bool op = f32_done[6 + m] || (onepass && DL.n_items < MFM_DW_MAXI && dw_onepass_supported(I, c.precision));
        if (op && !f32_done[6 + m]) DL.it[DL.n_items++] = I;
        dA_gemms(P, P->dec[m], P->dec_p[m], W, grads, tail, W + P->dec_init[m], P->dec_h[m], P->dec_h[m], true, op);
      }
    if (DL.n_items > 0) RUN(K_DEC_DW, dw_onepass_launch(DL, c.precision, s));
    // fp32 plans at small T*B: the chunked kernel (gemm_tn.hip: one load round trip per workgroup instead of a 20-step ring;
    // profiles/r02_gemm_tn.txt); MFM_GEMM_TN=0 / larger row counts / bf16 plans: the grouped GEMM
    long tn_rows = 1024;         // measured crossover: 640 rows 21.7 vs 24.8 us, 1280 rows equal, 2560 rows 67 vs 59 us
    if (const char* e = opt_get("MFM_GEMM_TN_MAXROWS")) tn_rows = atol(e);
    const bool tn_on = !c.precision && !(opt_get("MFM_GEMM_TN") && atoi(opt_get("MFM_GEMM_TN")) == 0);
    // bf16 plans: the products over B rows (the latent stack's 22 Linears on their fp32 records, the decoders' t = 0 products
    // on bf16-resident dA) go to the chunked fp32 kernel too -- 22 small outputs with K = B are all split-K prologue on the grouped kernel (48 us at
    // B = 2048) -- the rest (bf16-resident operands, sums over T*B rows) stays on the grouped bf16 GEMM
    // (bf16-resident plans: that is the whole tail, one launch either way -- B = 192 / 256 / 512 / 1024 / 2048: 9.8 vs 11.8,
    // 10.0 vs 12.9, 11.2 vs 16.2, 16.4 vs 23.8, 22.2 vs 48.5 us; fp32-stored bf16 plans, B < 192, keep the one grouped launch;
    // MFM_GEMM_TN_BF16_MINB moves the threshold)
    const long tn16_minb = opt_get("MFM_GEMM_TN_BF16_MINB") ? atol(opt_get("MFM_GEMM_TN_BF16_MINB")) : 192;
    if (c.precision && B >= tn16_minb && !(opt_get("MFM_GEMM_TN") && atoi(opt_get("MFM_GEMM_TN")) == 0)) {
      long tn_rows16 = 8192;
      if (const char* e = opt_get("MFM_GEMM_TN_MAXROWS_BF16")) tn_rows16 = atol(e);
      std::vector<MfmGemmDesc> small, rest;
      for (const MfmGemmDesc& d : tail)
        ((!d.c_bf16 && d.k <= tn_rows16 && d.k <= 4L * B && gemm_tn_supported(&d, 1, (int)tn_rows16, true)) ? small : rest).push_back(d);
      for (size_t done = 0; done < small.size(); done += MFM_TN_MAXP) {
        const int cnt = (int)std::min(small.size() - done, (size_t)MFM_TN_MAXP);
        RUN(K_ENC_DW, gemm_tn_launch(small.data() + done, cnt, (int)tn_rows16, true, s));
      }
      for (const MfmGemmDesc& d : rest) MFM_REQUIRE(!gemm_get_colsum_host(d), "plan: a product that carries bias column sums did not reach the chunked kernel");
      tail.swap(rest);
    }
    const int ntail = (int)tail.size();
    // the chunked kernel takes up to MFM_TN_MAXP products per launch (the MFN plans' ~90 in one), the grouped GEMM MFM_GEMM_MAXP
    const bool tail_tn = tn_on && ntail <= MFM_TN_MAXP && gemm_tn_supported(tail.data(), ntail, (int)std::min(tn_rows, (long)INT32_MAX), true);
    const int per = tail_tn ? MFM_TN_MAXP : MFM_GEMM_MAXP;
    for (int done = 0; done < ntail; done += per) {
      const int cnt = std::min(ntail - done, per);
      const MfmGemmDesc* td = tail.data() + done;
      // (the gradient buffer was cleared at the start of the step, so the tail's non-accumulating products may add)
      if (tn_on && gemm_tn_supported(td, cnt, (int)std::min(tn_rows, (long)INT32_MAX), true)) RUN(K_ENC_DW, gemm_tn_launch(td, cnt, (int)std::min(tn_rows, (long)INT32_MAX), true, s));
      else {
        for (int d2 = 0; d2 < cnt; d2 += MFM_GEMM_MAXP) {
          const int c2 = std::min(cnt - d2, (int)MFM_GEMM_MAXP);
          RUN(K_ENC_DW, c.precision ? mfm_gemm_grouped_bf16(td + d2, c2, s) : mfm_gemm_grouped_f32(td + d2, c2, s));
        }
      }
    }
  }
  return MFM_OK;
}

}  // namespace mfm

using namespace mfm;

extern "C" int mfm_plan_create(const MfmPlanConfig* cfg, const int64_t* param_offsets, int64_t n_params_total,
                               MfmPlan** out) {
  if (!cfg || !param_offsets || !out) { set_error("mfm_plan_create: null argument"); return MFM_ERR_ARG; }
  const MfmPlanConfig& c = *cfg;
  MFM_REQUIRE(c.T >= 1 && c.B >= 1, "plan: T=%d B=%d", c.T, c.B);
  MFM_REQUIRE(c.d_l > 0 && c.d_a > 0 && c.d_v > 0, "plan: input dims must be positive");
  MFM_REQUIRE(c.zl > 0 && c.za > 0 && c.zv > 0 && c.zy > 0 && c.fl > 0 && c.fa > 0 && c.fv > 0 && c.fy > 0,
              "plan: latent sizes must be positive");
  MFM_REQUIRE(c.output_dim >= 1 && c.output_dim <= 64, "plan: output_dim %d", c.output_dim);
  MFM_REQUIRE(c.loss_kind == 0 || c.loss_kind == 1, "plan: loss_kind %d", c.loss_kind);
  MFM_REQUIRE(c.precision == 0 || c.precision == 1, "plan: precision %d (0 = fp32, 1 = bf16 operands)", c.precision);
  MFM_REQUIRE(c.variant >= 0 && c.variant <= 2, "plan: variant %d (0 MFM_KL_EF, 1 MFM_KL, 2 MFM)", c.variant);
  if (c.variant != 0) {
    MFM_REQUIRE(c.hl > 0 && c.ha > 0 && c.hv > 0 && c.mem_dim > 0 && c.nn1 > 0 && c.nn2 > 0 && c.g1 > 0 && c.g2 > 0,
                "plan: MFN sizes must be positive (h_dims %d/%d/%d, memsize %d, NN1/NN2/gamma1/gamma2 %d/%d/%d/%d)",
                c.hl, c.ha, c.hv, c.mem_dim, c.nn1, c.nn2, c.g1, c.g2);
    MFM_REQUIRE(c.hl <= MFM_SEQ_MAX_RESIDENT_H && c.ha <= MFM_SEQ_MAX_RESIDENT_H && c.hv <= MFM_SEQ_MAX_RESIDENT_H,
                "plan: MFN LSTM sizes above %d are not supported on the fused plan (use the module path)", MFM_SEQ_MAX_RESIDENT_H);
    MFM_REQUIRE(2 * (c.hl + c.ha + c.hv) <= 1024, "plan: cStar width %d > 1024", 2 * (c.hl + c.ha + c.hv));
  }
  MfmPlan* P = new (std::nothrow) MfmPlan();
  if (!P) { set_error("plan: out of host memory"); return MFM_ERR_ARG; }
  P->cfg = c;
  P->pi = pidx_for(c.variant);
  P->gauss = nullptr;
  if (P->cfg.reg_scale == 0.0f) P->cfg.reg_scale = 1.0f;
  for (int i = 0; i < P->pi.count; ++i) {
    P->off[i] = param_offsets[i];
    if (param_offsets[i] < 0 || param_offsets[i] >= n_params_total) {
      delete P; set_error("plan: param offset %d out of range", i); return MFM_ERR_ARG;
    }
  }
  P->n_params = n_params_total;
  P->timing_mask = 0; P->timing_every = 1; P->pool_used = 0; P->calls = 0; P->grads_prezeroed = nullptr;
  // MFM_SHARED_DEVICE=1 (several ranks / processes drive this GPU): the default of the "handover" option for plans created
  // from now on; the host side sets the option itself where it can tell (train.py::_mark_shared_device)
  if (const char* e = opt_get("MFM_SHARED_DEVICE")) P->opt_handover = atoi(e) == 0;
  if (const char* e = opt_get("MFM_BF16_DOT")) P->opt_bf16_dot = atoi(e) != 0;
  int rc = build(P);
  if (rc != MFM_OK) { delete P; return rc; }
  *out = P;
  return MFM_OK;
}

extern "C" int mfm_plan_num_params(int32_t variant) {
  return (variant >= 0 && variant <= 2) ? pidx_for(variant).count : 0;
}

extern "C" int mfm_plan_set_gauss(MfmPlan* P, const float* gauss) {
  if (!P) { set_error("mfm_plan_set_gauss: null plan"); return MFM_ERR_ARG; }
  P->gauss = gauss;
  return MFM_OK;
}

extern "C" void mfm_plan_destroy(MfmPlan* P) {
  if (!P) return;
  for (auto& t : P->pool) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
  opt_table_free(P->opts);
  if (P->host_status) (void)hipHostFree(P->host_status);
  delete P;
}

extern "C" int64_t mfm_plan_debug_offset(const MfmPlan* P) { return P ? P->dbg_off * (int64_t)sizeof(float) : -1; }

extern "C" int64_t mfm_plan_workspace_bytes(const MfmPlan* P) { return P ? P->ws_floats * (int64_t)sizeof(float) : 0; }

extern "C" int mfm_plan_init_workspace(MfmPlan* P, void* workspace, void* stream) {
  if (!P || !workspace) { set_error("mfm_plan_init_workspace: null argument"); return MFM_ERR_ARG; }
  float* W = (float*)workspace;
  hipStream_t s = (hipStream_t)stream;
  if (!P->host_status) {           // (not in mfm_plan_create: that one stays free of the HIP runtime)
    void* hp = nullptr;
    MFM_HIP_CHECK(hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocCoherent));
    memset(hp, 0, 64);
    P->host_status = reinterpret_cast<unsigned*>(hp);
  }
  MFM_HIP_CHECK(hipMemsetAsync(W, 0, (size_t)P->ws_floats * sizeof(float), s));
  MFM_HIP_CHECK(hipMemcpyAsync(W + P->lat_ops_off, P->lat_ops, sizeof(P->lat_ops), hipMemcpyHostToDevice, s));
  MFM_HIP_CHECK(hipMemcpyAsync(W + P->lat_items_off, P->lat_items.data(), P->lat_items.size() * sizeof(int),
                               hipMemcpyHostToDevice, s));
  return fill_launch(W + P->ones, (int64_t)P->T * P->B, 1.0f, s);
}

extern "C" int mfm_plan_set_option(MfmPlan* P, const char* key, int64_t value) {
  if (!P || !key) { set_error("mfm_plan_set_option: null argument"); return MFM_ERR_ARG; }
  if (!strcmp(key, "handover")) {
    P->opt_handover = value != 0;
    // (the launch-form states restart: "in use" must describe the launches the next call issues -- mfm_plan_kernel_flops)
    if (P->projfold_state == 1 || !P->opt_handover) P->projfold_state = 0;
    if (P->dwfold_state == 1 || !P->opt_handover) P->dwfold_state = 0;
  }
  else if (!strcmp(key, "handover_timeout_us")) { MFM_REQUIRE(value >= 1 && value <= 10000000, "mfm_plan_set_option: handover_timeout_us %lld", (long long)value); P->opt_timeout_us = value; }
  else if (!strcmp(key, "grad_guard_offset")) { MFM_REQUIRE(value >= -1 && value < P->n_params, "mfm_plan_set_option: grad_guard_offset %lld outside the buffer of %lld elements", (long long)value, (long long)P->n_params); P->opt_guard = value; }
  else if (!strcmp(key, "bf16_dot")) P->opt_bf16_dot = value != 0;
  else if (!strcmp(key, "inject_fault")) { MFM_REQUIRE(value >= 0 && value <= 2, "mfm_plan_set_option: inject_fault %lld", (long long)value); P->opt_fault = (int)value; }
  else { set_error("mfm_plan_set_option: unknown key '%s'", key); return MFM_ERR_ARG; }
  return MFM_OK;
}
extern "C" int mfm_plan_set_option_str(MfmPlan* P, const char* key, const char* value) {
  if (!P || !key) { set_error("mfm_plan_set_option_str: null argument"); return MFM_ERR_ARG; }
  MFM_REQUIRE(strncmp(key, "MFM_", 4) == 0, "mfm_plan_set_option_str: '%s' is not an MFM_* switch", key);
  opt_table_set(P->opts, key, value);
  // launch forms that were decided by trying (fold launches, role workgroups) are tried again under the new switches
  P->fold_state = 0; P->projfold_state = 0; P->dwfold_state = 0; P->dw_table_key = -1;
  return MFM_OK;
}
extern "C" int mfm_plan_get_option(const MfmPlan* P, const char* key, int64_t* value) {
  if (!P || !key || !value) { set_error("mfm_plan_get_option: null argument"); return MFM_ERR_ARG; }
  if (!strcmp(key, "handover")) *value = P->opt_handover;
  else if (!strcmp(key, "handover_timeout_us")) *value = P->opt_timeout_us;
  else if (!strcmp(key, "grad_guard_offset")) *value = P->opt_guard;
  else if (!strcmp(key, "inject_fault")) *value = P->opt_fault;
  else if (!strcmp(key, "bf16_dot")) *value = P->opt_bf16_dot;
  // read-only: whether the last forward / backward of the plan ran on role workgroups
  else if (!strcmp(key, "proj_roles_active")) *value = P->projfold_state == 1;
  else if (!strcmp(key, "dw_roles_active")) *value = P->dwfold_state == 1;
  else { set_error("mfm_plan_get_option: unknown key '%s'", key); return MFM_ERR_ARG; }
  return MFM_OK;
}
extern "C" int mfm_plan_state_layout(const MfmPlan* P, int64_t* out) {
  if (!P || !out) { set_error("mfm_plan_state_layout: null argument"); return MFM_ERR_ARG; }
  for (int i = 0; i < 8; ++i) out[i] = 0;
  const int64_t f = (int64_t)sizeof(float);
  out[0] = P->losses * f; out[1] = (P->losses + MfmPlan::ST_STATUS) * f; out[2] = (P->losses + MfmPlan::ST_TICK) * f;
  out[3] = (P->losses + MfmPlan::ST_DW_TICK) * f;
  return MFM_OK;
}
extern "C" int mfm_plan_host_status(MfmPlan* P, uint32_t** out) {
  if (!P || !out) { set_error("mfm_plan_host_status: null argument"); return MFM_ERR_ARG; }
  *out = P->host_status;
  return MFM_OK;
}

extern "C" int mfm_plan_clear_status(MfmPlan* P, void* workspace, void* stream) {
  if (!P || !workspace) { set_error("mfm_plan_clear_status: null argument"); return MFM_ERR_ARG; }
  if (P->host_status) { P->host_status[0] = 0; P->host_status[1] = 0; }
  MFM_HIP_CHECK(hipMemsetAsync(P->status_ptr((float*)workspace), 0, sizeof(unsigned), (hipStream_t)stream));
  return MFM_OK;
}

extern "C" int mfm_plan_forward(MfmPlan* P, const float* params, const float* x, const void* y, int train,
                                uint64_t seed, void* workspace, float* xhat_l, float* xhat_a, float* xhat_v,
                                float* y_hat, float* losses, void* stream) {
  if (!P || !params || !x || !workspace) { set_error("mfm_plan_forward: null argument"); return MFM_ERR_ARG; }
  float* xo[3] = {xhat_l, xhat_a, xhat_v};
  return forward(P, params, x, y, train, seed, (float*)workspace, xo, y_hat, losses, (hipStream_t)stream);
}

extern "C" int mfm_plan_backward(MfmPlan* P, const float* params, const float* x, const void* y, int stage,
                                 void* workspace, float* grads, void* stream) {
  if (!P || !params || !x || !workspace || !grads) { set_error("mfm_plan_backward: null argument"); return MFM_ERR_ARG; }
  MFM_REQUIRE(stage >= 0 && stage <= 2, "mfm_plan_backward: stage %d", stage);
  MFM_REQUIRE(y || stage == 1, "mfm_plan_backward: labels required unless stage==1");
  return backward(P, params, x, y, stage, (float*)workspace, grads, (hipStream_t)stream);
}

extern "C" int mfm_plan_forward_train(MfmPlan* P, const float* params, const float* x, uint64_t seed, void* workspace,
                                      float* grads_to_zero, void* stream) {
  if (!P || !params || !x || !workspace) { set_error("mfm_plan_forward_train: null argument"); return MFM_ERR_ARG; }
  float* xo[3] = {nullptr, nullptr, nullptr};
  return forward(P, params, x, nullptr, 1, seed, (float*)workspace, xo, nullptr, nullptr, (hipStream_t)stream, grads_to_zero);
}

extern "C" int mfm_plan_out_layout(const MfmPlan* P, int64_t* out) {
  if (!P || !out) { set_error("mfm_plan_out_layout: null argument"); return MFM_ERR_ARG; }
  for (int i = 0; i < 8; ++i) out[i] = -1;
  if (!P->st16)
    for (int m = 0; m < 3; ++m) out[m] = P->xhat[m] * (int64_t)sizeof(float);
  out[3] = P->yhat * (int64_t)sizeof(float);
  return MFM_OK;
}

extern "C" int mfm_plan_backward_weighted(MfmPlan* P, const float* params, const float* x, const void* y,
                                          const MfmLossWeights* w, void* workspace, float* grads, void* stream) {
  if (!P || !params || !x || !workspace || !grads || !w) { set_error("mfm_plan_backward_weighted: null argument"); return MFM_ERR_ARG; }
  const MfmPlanConfig& c = P->cfg;
  const bool gen_on = w->gen_l != 0.0f || w->gen_a != 0.0f || w->gen_v != 0.0f;
  // the forward's fc1 epilogue left d x_hat_m = 2 lda_m (x_hat_m - x_m) / count_m behind: the reconstruction weights must be
  // the plan's own, or all zero
  if (gen_on && (w->gen_l != c.lda_xl || w->gen_a != c.lda_xa || w->gen_v != c.lda_xv)) {
    set_error("mfm_plan_backward_weighted: reconstruction weights (%g, %g, %g) differ from the plan's (%g, %g, %g)", w->gen_l,
              w->gen_a, w->gen_v, c.lda_xl, c.lda_xa, c.lda_xv);
    return MFM_ERR_UNSUPPORTED;
  }
  MFM_REQUIRE(y || (w->disc == 0.0f && !w->write_disc_loss), "mfm_plan_backward_weighted: labels required for the discriminative term");
  MFM_REQUIRE(!P->st16, "mfm_plan_backward_weighted: not available on a bf16-resident plan");
  LossW lw{w->disc, gen_on ? 1 : 0, w->reg, w->write_disc_loss};
  return backward(P, params, x, y, 0, (float*)workspace, grads, (hipStream_t)stream, nullptr, &lw);
}

extern "C" int mfm_plan_backward_ext(MfmPlan* P, const float* params, const float* x, const float* d_xhat_l,
                                     const float* d_xhat_a, const float* d_xhat_v, const float* d_yhat,
                                     const float* d_reg, void* workspace, float* grads, void* stream) {
  if (!P || !params || !x || !workspace || !grads || !d_xhat_l || !d_xhat_a || !d_xhat_v || !d_yhat || !d_reg) {
    set_error("mfm_plan_backward_ext: null argument");
    return MFM_ERR_ARG;
  }
  ExtGrads ext;
  ext.d_xhat[0] = d_xhat_l; ext.d_xhat[1] = d_xhat_a; ext.d_xhat[2] = d_xhat_v;
  ext.d_yhat = d_yhat; ext.d_reg = d_reg;
  return backward(P, params, x, nullptr, 0, (float*)workspace, grads, (hipStream_t)stream, &ext);
}

extern "C" int mfm_plan_grad_step(MfmPlan* P, const float* params, float* grads, const float* x, const void* y,
                                  uint64_t seed, void* workspace, float* losses, void* stream) {
  if (!P || !params || !grads || !x || !y || !workspace) {
    set_error("mfm_plan_grad_step: null argument");
    return MFM_ERR_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  float* xo[3] = {nullptr, nullptr, nullptr};
  int rc = forward(P, params, x, y, 1, seed, (float*)workspace, xo, nullptr, losses, s, grads);
  if (rc != MFM_OK) return rc;
  return backward(P, params, x, y, 0, (float*)workspace, grads, s);
}

extern "C" int mfm_plan_train_step(MfmPlan* P, float* params, float* grads, float* adam_m, float* adam_v,
                                   const float* x, const void* y, uint64_t seed, int32_t step, float lr,
                                   float grad_scale, void* workspace, float* losses, void* stream) {
  if (!P || !params || !grads || !adam_m || !adam_v || !x || !y || !workspace) {
    set_error("mfm_plan_train_step: null argument");
    return MFM_ERR_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  float* xo[3] = {nullptr, nullptr, nullptr};
  int rc = forward(P, params, x, y, 1, seed, (float*)workspace, xo, nullptr, losses, s, grads);
  if (rc != MFM_OK) return rc;
  // (Adam as tail workgroups of the weight-gradient launch was built and measured: slower, and its inactive code cost every
  // grouped GEMM launch ~1.5 us -- removed again, profiles/r02_adam_tail.txt)
  rc = backward(P, params, x, y, 0, (float*)workspace, grads, s);
  if (rc != MFM_OK) return rc;
  const float* guard = (P->opt_guard >= 0 && P->opt_guard < P->n_params) ? grads + P->opt_guard : nullptr;
  RUN(K_ADAM, adam_launch(params, grads, adam_m, adam_v, P->n_params, step, lr, 0.9f, 0.999f, 1e-8f, grad_scale, s, guard));
  return MFM_OK;
}

extern "C" int mfm_plan_train_step_staged(MfmPlan* P, float* params, float* grads, float* adam_m, float* adam_v,
                                          const float* x, const void* y, uint64_t seed, int32_t stage,
                                          const MfmAdamSpan* spans, int32_t nspans, float lr, float grad_scale,
                                          void* workspace, float* losses, void* stream) {
  if (!P || !params || !grads || !adam_m || !adam_v || !x || !y || !workspace || !spans) {
    set_error("mfm_plan_train_step_staged: null argument");
    return MFM_ERR_ARG;
  }
  MFM_REQUIRE(stage >= 0 && stage <= 2, "mfm_plan_train_step_staged: stage %d", stage);
  hipStream_t s = (hipStream_t)stream;
  float* xo[3] = {nullptr, nullptr, nullptr};
  int rc = forward(P, params, x, y, 1, seed, (float*)workspace, xo, nullptr, losses, s, grads);
  if (rc != MFM_OK) return rc;
  rc = backward(P, params, x, y, stage, (float*)workspace, grads, s);
  if (rc != MFM_OK) return rc;
  const float* guard = (P->opt_guard >= 0 && P->opt_guard < P->n_params) ? grads + P->opt_guard : nullptr;
  RUN(K_ADAM, adam_spans_launch(params, grads, adam_m, adam_v, spans, nspans, lr, 0.9f, 0.999f, 1e-8f, grad_scale, s, guard));
  return MFM_OK;
}

extern "C" int mfm_plan_latent_layout(const MfmPlan* P, int64_t* out) {
  if (!P || !out) { set_error("mfm_plan_latent_layout: null argument"); return MFM_ERR_ARG; }
  for (int i = 0; i < 32; ++i) out[i] = 0;
  out[0] = P->lat_rec * (int64_t)sizeof(float);
  out[1] = P->lat_grd * (int64_t)sizeof(float);
  out[2] = P->lat.rec_size;
  const int order[4] = {0, 1, 2, 3};      // l, a, v, y
  for (int i = 0; i < 4; ++i) {
    const int e = order[i];
    out[3 + i] = P->lay_m1[e];
    out[8 + i] = P->lay_f1[e];
    out[13 + i] = P->lat.f_n[e];
    out[17 + i] = P->lat.f_off[e];
    out[23 + i] = P->lat.mu_off[e];
    out[27 + i] = P->lat.z_n[e];
  }
  out[7] = P->lay_mc;
  out[12] = P->lay_c1;
  out[21] = P->lat.yhat_off;
  out[22] = P->lat.row_path;
  return MFM_OK;
}

extern "C" int mfm_plan_seq_layout(const MfmPlan* P, int32_t which, int64_t* out) {
  if (!P || !out) { set_error("mfm_plan_seq_layout: null argument"); return MFM_ERR_ARG; }
  MFM_REQUIRE(which >= 0 && which < P->n_enc + 3, "mfm_plan_seq_layout: LSTM %d of %d", which, P->n_enc + 3);
  for (int i = 0; i < 12; ++i) out[i] = 0;
  const bool dec = which >= P->n_enc;
  const mfm::SeqBuf& sb = dec ? P->dec[which - P->n_enc] : P->enc[which];
  const int64_t f = (int64_t)sizeof(float);
  out[0] = sb.gates * f; out[1] = sb.hs * f; out[2] = sb.cs * f; out[3] = sb.h; out[4] = sb.Hp; out[5] = P->st16 ? 1 : 0;
  out[6] = dec ? 1 : 0;
  if (dec) {
    const int m = which - P->n_enc;
    out[7] = P->dec_dhs[m] * f; out[8] = P->dxhat[m] * f; out[9] = P->dxh_ld[m]; out[10] = P->dec_d[m];
  } else {
    out[7] = P->h_last[which] >= 0 ? P->h_last[which] * f : -1;
  }
  out[11] = P->seq_bf16 ? 1 : 0;
  return MFM_OK;
}

extern "C" int mfm_plan_mfn_layout(const MfmPlan* P, int64_t* out) {
  if (!P || !out) { set_error("mfm_plan_mfn_layout: null argument"); return MFM_ERR_ARG; }
  for (int i = 0; i < 16; ++i) out[i] = 0;
  if (P->cfg.variant == 0) { set_error("mfm_plan_mfn_layout: the plan has no Memory Fusion Network (variant 0)"); return MFM_ERR_ARG; }
  const int64_t f = (int64_t)sizeof(float);
  out[0] = P->cstar * f; out[1] = P->h1 * f; out[2] = P->m1 * f; out[3] = P->att * f; out[4] = P->attended * f;
  out[5] = P->h2 * f; out[6] = P->m2 * f; out[7] = P->chat * f; out[8] = P->mem_out * f; out[9] = P->zyin * f;
  out[10] = P->A2; out[11] = P->cfg.nn1; out[12] = P->cfg.nn2; out[13] = P->cfg.mem_dim; out[14] = P->nzy;
  out[15] = (int64_t)P->T * P->B;
  return MFM_OK;
}

// ---- timing: HIP events on the launch stream around the kernels selected by `mask`
extern "C" int mfm_plan_set_timing(MfmPlan* P, int mask) {
  if (!P) return MFM_ERR_ARG;
  P->timing_mask = mask;
  return MFM_OK;
}
extern "C" int mfm_plan_set_timing_every(MfmPlan* P, int every) {
  if (!P || every < 1) return MFM_ERR_ARG;
  P->timing_every = every;
  return MFM_OK;
}
extern "C" int mfm_plan_num_kernels(void) { return K_COUNT; }
extern "C" const char* mfm_plan_kernel_name(int kid) {
  static const char* names[K_COUNT] = {"proj_gemm", "enc_seq_fwd", "latent_fwd", "dec_seq_fwd", "fc1_mse_gemm", "mse",
                                       "fc1_bwd_gemm", "dec_seq_bwd", "lstm_dw_onepass", "latent_bwd", "enc_seq_bwd",
                                       "dw_gemm", "adam", "latent_dw_gemm", "bf16_weight_pack", "mfn_glue", "mfn_att_fwd_gemm",
                                       "mfn_mem_fwd", "mfn_heads_gemm", "mfn_mem_bwd", "mfn_att_bwd_gemm", "mmd"};
  return (kid >= 0 && kid < K_COUNT) ? names[kid] : "?";
}
// Synchronises on the recorded events, adds elapsed ms / launch counts per kernel id, resets the pool.
extern "C" int mfm_plan_collect_timing(MfmPlan* P, double* sum_ms /*[K_COUNT]*/, int64_t* count /*[K_COUNT]*/) {
  if (!P || !sum_ms || !count) return MFM_ERR_ARG;
  for (int i = 0; i < K_COUNT; ++i) { sum_ms[i] = 0.0; count[i] = 0; }
  for (size_t i = 0; i < P->pool_used; ++i) {
    TimingPair& t = P->pool[i];
    MFM_HIP_CHECK(hipEventSynchronize(t.b));
    float ms = 0.0f;
    MFM_HIP_CHECK(hipEventElapsedTime(&ms, t.a, t.b));
    if (t.kid >= 0 && t.kid < K_COUNT) { sum_ms[t.kid] += ms; count[t.kid]++; }
  }
  P->pool_used = 0;
  return MFM_OK;
}

// Cost of one event bracket with nothing inside it (two hipEventRecord on `stream`): the median of 33 empty
// brackets.  A bracket around a kernel reads kernel duration + about this much (the records are packets of
// their own); bench.py subtracts it so that its per-kernel time can be compared with rocprofv3's.
extern "C" int mfm_timing_bracket_overhead_ms(void* stream, double* ms_out) {
  if (!ms_out) return MFM_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  constexpr int N = 33;
  hipEvent_t a[N], b[N];
  for (int i = 0; i < N; ++i) { MFM_HIP_CHECK(hipEventCreate(&a[i])); MFM_HIP_CHECK(hipEventCreate(&b[i])); }
  for (int i = 0; i < N; ++i) { MFM_HIP_CHECK(hipEventRecord(a[i], s)); MFM_HIP_CHECK(hipEventRecord(b[i], s)); }
  MFM_HIP_CHECK(hipEventSynchronize(b[N - 1]));
  float v[N];
  for (int i = 0; i < N; ++i) MFM_HIP_CHECK(hipEventElapsedTime(&v[i], a[i], b[i]));
  std::sort(v, v + N);
  *ms_out = v[N / 2];
  for (int i = 0; i < N; ++i) { (void)hipEventDestroy(a[i]); (void)hipEventDestroy(b[i]); }
  return MFM_OK;
}

// ---- algorithmic work: 2*4h*(d+h) per cell step, 2*in*out per Linear
static double fwd_flops_per_sample(const MfmPlan* P) {
  const MfmPlanConfig& c = P->cfg;
  double f = 0.0;
  for (int e = 0; e < P->n_enc; ++e) {
    const double h = P->enc_h[e], d = P->enc_d[e];
    f += P->T * 2.0 * 4.0 * h * (d + h);
  }
  for (int m = 0; m < 3; ++m) {
    const double h = P->dec_h[m], d = P->dec_d[m];
    f += P->T * (2.0 * 4.0 * h * (h + h) + 2.0 * h * d);
  }
  for (int i = 0; i < P->lat.nops; ++i) f += 2.0 * P->lat_ops[i].K * P->lat_ops[i].N;
  if (c.variant != 0) {      // MFN per time step: attention (4 Linears), the two gamma nets; once: heads on mfn_last
    const double A2 = P->A2, M = c.mem_dim;
    f += P->T * 2.0 * (A2 * c.nn1 + c.nn1 * A2 + A2 * c.nn2 + c.nn2 * M + (A2 + M) * (c.g1 + c.g2) + (c.g1 + c.g2) * M);
    f += 2.0 * (P->tot + M) * P->nzy;
  }
  return f;
}
extern "C" double mfm_plan_flops_per_step(const MfmPlan* P) { return P ? 3.0 * fwd_flops_per_sample(P) * P->B : 0.0; }
extern "C" double mfm_plan_bytes_per_step(const MfmPlan* P) {
  if (!P) return 0.0;
  double sh = 0.0;
  for (int e = 0; e < P->n_enc; ++e) sh += P->enc_h[e];
  for (int m = 0; m < 3; ++m) sh += P->dec_h[m];
  const double per_sample = 2.0 * P->T * P->D * 4.0 + 4.0 + 2.0 * P->T * 6.0 * sh * 4.0;
  return per_sample * P->B + 10.0 * (double)P->n_params * 4.0;
}
// Algorithmic FLOPs of ONE launch of kernel `kid` (recurrent/GEMM kernels only; 0 otherwise).
extern "C" double mfm_plan_kernel_flops(const MfmPlan* P, int kid) {
  if (!P) return 0.0;
  OptScope _opts(P->opts);
  const double TB = (double)P->T * P->B;
  double f = 0.0;
  switch (kid) {
    case K_PROJ: for (int e = 0; e < P->n_enc; ++e) f += TB * 2.0 * 4.0 * P->enc_h[e] * P->enc_d[e]; break;
    // (variants 1, 2 run the six encoder recurrences as two launches: this is the sum of both)
    case K_ENC_FWD: case K_ENC_BWD:
      for (int e = 0; e < P->n_enc; ++e) f += TB * 2.0 * 4.0 * P->enc_h[e] * P->enc_h[e];
      if (P->fold_state == 1)          // fold launches: the rows' latent chains run in the same workgroups
        for (int i = 0; i < P->lat.nops; ++i) f += 2.0 * P->B * P->lat_ops[i].N * P->lat_ops[i].K;
      // role workgroups (B <= 32): the forward launch also produces the input projections, the backward launch every
      // weight gradient of the step (proj_role_dev.h, dw_role_dev.h)
      if (kid == K_ENC_FWD && P->projfold_state == 1) f += mfm_plan_kernel_flops(P, K_PROJ);
      if (kid == K_ENC_BWD && P->dwfold_state == 1) {
        for (int e = 0; e < P->n_enc; ++e) f += TB * 2.0 * 4.0 * P->enc_h[e] * (P->enc_d[e] + P->enc_h[e]);
        for (int m = 0; m < 3; ++m) f += (TB - P->B) * 2.0 * 4.0 * P->dec_h[m] * P->dec_h[m];
        for (int m = 0; m < 3; ++m) f += TB * 2.0 * P->dec_h[m] * P->dec_d[m];                      // dWfc
        for (int m = 0; m < 3; ++m) f += (double)P->B * 2.0 * 4.0 * P->dec_h[m] * P->dec_h[m];      // the decoders' t = 0 product
        for (int i = 0; i < P->lat.nops; ++i) f += 2.0 * P->B * P->lat_ops[i].N * P->lat_ops[i].K;  // latent dW
      }
      break;
    case K_DEC_FWD: case K_DEC_BWD: for (int m = 0; m < 3; ++m) f += TB * 2.0 * 4.0 * P->dec_h[m] * P->dec_h[m]; break;
    case K_FC1_FWD: {
      for (int m = 0; m < 3; ++m) f += TB * 2.0 * P->dec_h[m] * P->dec_d[m];
      // the fused kernel (dec_fc1.hip; the plan's default up to 5120 rows) also forms dH = dx_hat Wfc in the same launch
      long fc1_max_rows = 5120;
      if (const char* e = opt_get("MFM_FC1_FUSED_MAXROWS")) fc1_max_rows = atol(e);
      const bool on = !(opt_get("MFM_FC1_FUSED") && atoi(opt_get("MFM_FC1_FUSED")) == 0);
      if (on && TB <= (double)fc1_max_rows) f *= 2.0;
      break;
    }
    case K_FC1_BWD: for (int m = 0; m < 3; ++m) f += TB * 2.0 * P->dec_h[m] * P->dec_d[m]; break;   // dH only
    // weight gradients: one grouped launch (K_ENC_DW) -- except on bf16-resident plans, where the sums over the T*B rows
    // (LSTMs, decoder fc1) run on the one-pass kernel (K_DEC_DW) and K_ENC_DW keeps the B-row products
    case K_DEC_DW: case K_ENC_DW: {
      double rows_f = 0.0, rest = 0.0;
      for (int e = 0; e < P->n_enc; ++e) rows_f += TB * 2.0 * 4.0 * P->enc_h[e] * (P->enc_d[e] + P->enc_h[e]);
      for (int m = 0; m < 3; ++m) rows_f += (TB - P->B) * 2.0 * 4.0 * P->dec_h[m] * P->dec_h[m];
      for (int m = 0; m < 3; ++m) rows_f += TB * 2.0 * P->dec_h[m] * P->dec_d[m];            // dWfc
      for (int m = 0; m < 3; ++m) rest += (double)P->B * 2.0 * 4.0 * P->dec_h[m] * P->dec_h[m];      // the decoders' t = 0 product
      for (int i = 0; i < P->lat.nops; ++i) rest += 2.0 * P->B * P->lat_ops[i].N * P->lat_ops[i].K;   // latent dW
      if (P->st16) f = (kid == K_DEC_DW) ? rows_f : rest;
      else f = (kid == K_ENC_DW) ? rows_f + rest : 0.0;
      break;
    }
    // Memory Fusion Network (variants 1, 2), GEMM form: per-launch AVERAGE over the launches that share the timer id
    case K_MFN_ATT_FWD: {
      const MfmPlanConfig& c = P->cfg;
      const double A2 = P->A2, M = c.mem_dim;
      f = TB * 2.0 * (A2 * c.nn1 + c.nn1 * A2 + A2 * (c.nn2 + c.g1 + c.g2) + c.nn2 * M) / 4.0;
      break;
    }
    case K_MFN_ATT_BWD: {
      const MfmPlanConfig& c = P->cfg;
      const double A2 = P->A2, M = c.mem_dim;
      f = TB * 2.0 * (M * c.nn2 + (c.nn2 + c.g1 + c.g2) * A2 + A2 * c.nn1 + c.nn1 * A2) / 4.0;
      break;
    }
    case K_MFN_MEM_FWD: case K_MFN_MEM_BWD: {
      const MfmPlanConfig& c = P->cfg;
      f = TB * 2.0 * 2.0 * c.mem_dim * (c.g1 + c.g2);        // W_mem,n mem and W_fc2,n u_n for both gates
      break;
    }
    case K_MFN_HEADS: f = 2.0 * P->B * (P->tot + P->cfg.mem_dim) * P->cfg.zy * (P->cfg.variant == 1 ? 2 : 1); break;
    default: break;
  }
  return f;
}
