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
//
// This file: the forward chain F0..F4 (mfm::forward) and the helpers both chains share; plan_backward.hip holds B0..B5,
// plan_mfn.hip the Memory Fusion Network's launches, plan_build.hip the tables, plan.hip the C ABI.
#include "plan_internal.h"

namespace mfm {

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
bool stream_capturing(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
  return st == hipStreamCaptureStatusActive;
}

MfmSeqDesc seq_desc(const MfmPlan* P, const SeqBuf& sb, int pbase, const float* params, float* W, bool dec) {
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


int forward(MfmPlan* P, const float* params, const float* x, const void* y, int train, uint64_t seed,
                   float* W, float* xhat_out[3], float* yhat_out, float* losses_out, hipStream_t s,
                   float* grads_to_zero) {
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
  WtImgItem wt_items[MFM_IMG_MAX];
  int n_wt_items = 0, n_wf_items = 0;
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
        I.fwd = 0;
      }
      // launches without projection role workgroups (MFM_KL / MFM, B > 32): the decoders' forward-order images (lstm_seq_dev.h)
      // ride on the same image-writer blocks, behind the transposed ones
      if (T >= 2 && !(opt_get("MFM_WF_IMG") && atoi(opt_get("MFM_WF_IMG")) == 0) && (long)3 * B < 6L * device_cus()) {
        bool ok = true;
        for (int k = 0; k < 6; ++k) ok = ok && P->wf_img[k] >= 0;
        for (int k = 0; k < 6 && ok; ++k) {
          const int m = k % 3;
          const bool sum = k >= 3;
          WtImgItem& I = wt_items[n_wt_items + n_wf_items++];
          const int pb = P->dec_p[m];
          I.w_hh = sum ? params + P->off[pb + W_HH] : nullptr; I.w_ih = params + P->off[pb + W_IH];
          I.img = W + P->wf_img[sum ? m : 3 + m]; I.h = P->dec[m].h;
          I.HKB = round_up(4 * round_up(cdiv(P->dec[m].h, 4), 2), 16);
          I.fwd = 1;
        }
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
    // the decoders' steps >= 1 weights in the forward's register order: their launch reloads them with coalesced loads
    if (!(opt_get("MFM_WF_IMG") && atoi(opt_get("MFM_WF_IMG")) == 0)) {
      for (int k = 0; k < 6; ++k) {           // W_ih (the step-0 weights) first: the decoder launch asks for them first
        const int m = k % 3;
        const bool sum = k >= 3;
        if (P->wf_img[m] < 0 || P->wf_img[3 + m] < 0 || T < 2) continue;
        WtImgItem& I = PR.wf[PR.n_wf++];
        const int pb = P->dec_p[m];
        I.w_hh = sum ? params + P->off[pb + W_HH] : nullptr; I.w_ih = params + P->off[pb + W_IH];
        I.img = W + P->wf_img[sum ? m : 3 + m]; I.h = P->dec[m].h;
        I.HKB = round_up(4 * round_up(cdiv(P->dec[m].h, 4), 2), 16);
      }
    }
    PR.zs.ptr[0] = nullptr; PR.zs.n[0] = 0;
    for (int e = 0; e < 4; ++e) {
      const int pb = P->enc_p[e];
      PR.e[e].w = params + P->off[pb + W_IH]; PR.e[e].b_ih = params + P->off[pb + B_IH]; PR.e[e].b_hh = params + P->off[pb + B_HH];
      PR.e[e].k_off = P->enc_xoff[e]; PR.e[e].k = P->enc_d[e];
    }
    // the chains' stages behind the z -> f MLPs (classifier, logvar heads, losses, y_hat) are left to tail blocks of the decoder
    // launch when that launch can carry them (latent_row_dev.h, mode 3)
    {
      const int dech[3] = {P->dec[0].h, P->dec[1].h, P->dec[2].h};
      PR.lat_tail = seq_small_dectail_supported(T, B, dech, L) ? 1 : 0;
    }
    int rc;
    { Timer _t(P, s, K_ENC_FWD); rc = seq_foldproj_launch(q, 4, T, B, L, params, PR, s); }
    if (rc == MFM_OK) { folded = true; P->projfold_state = 1; P->fold_state = 1; P->ever_handover = true; if (PR.n_wt) P->wt_call = P->calls;
                        if (PR.n_wf == 6) P->wf_call = P->calls;
                        if (PR.lat_tail) P->lat_tail_call = P->calls; }
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
    if (P->fold_state == 1) { Timer _t(P, s, K_ENC_FWD); rc = seq_fold_launch(q, 4, T, B, false, L, params, nullptr, s, nullptr, n_wt_items ? wt_items : nullptr, n_wt_items + n_wf_items, &wrote); }
    else rc = seq_fold_launch(q, 4, T, B, false, L, params, nullptr, s, nullptr, n_wt_items ? wt_items : nullptr, n_wt_items + n_wf_items, &wrote);
    if (rc == MFM_OK) { folded = true; P->fold_state = 1; if (wrote) { P->wt_call = P->calls; if (n_wf_items == 6) P->wf_call = P->calls; } }
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
      RUN(K_ENC_FWD, seq_fwd_img_launch(q, n, T, B, wt_items, n_wt_items + n_wf_items, &wrote, s));
      if (wrote) { P->wt_call = P->calls; if (n_wf_items == 6) P->wf_call = P->calls; }
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
    const bool wf_on = !seq_bf16 && P->wf_call == P->calls;
    const float* wf[6];
    for (int k = 0; k < 6; ++k) wf[k] = wf_on ? W + P->wf_img[k] : nullptr;
    bool dec_done = false;
    if (P->lat_tail_call == P->calls) {          // the encoder launch left the chains' tails behind: they ride on this launch
      int rc;
      { Timer _t(P, s, K_DEC_FWD); rc = seq_dec_tail_launch(q, 3, T, B, wf_on ? wf : nullptr, L, params, s); }
      if (rc == MFM_OK) dec_done = true;
      else if (rc == MFM_ERR_UNSUPPORTED) RUN(K_LAT_FWD, latent_fwd_tail_launch(L, params, s));      // (a launch of their own)
      else return rc;
    }
    if (dec_done) {}
    else if (wf_on) RUN(K_DEC_FWD, seq_fwd_wf_launch(q, 3, T, B, wf, s));
    else RUN(K_DEC_FWD, seq_bf16 ? mfm_lstm_seq_fwd_bf16(q, 3, T, B, s) : mfm_lstm_seq_fwd(q, 3, T, B, s));
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
    MFM_LAUNCH_TIMED(tick_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<unsigned long long*>(P->tick_ptr(W)), (unsigned*)nullptr);
    MFM_LAUNCH_CHECK("tick_kernel");
  }
  return MFM_OK;
}

// `only_init`: just the decoders' t = 0 input product (the rest went to the one-pass kernel of dw_bf16.hip)

}  // namespace mfm
