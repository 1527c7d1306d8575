// The fused plan, part 1: what is decided ONCE per (model sizes, T, B) -- the index of every parameter tensor, the workspace
// carving (saved activations, records, flags, images), the latent stack's op table and the row kernels' item tables.
// (reference: the wiring of MFM_KL_EF / MFM_KL / MFM, mfm_model.py:557-660 / 662-764 / 469-555)
#include "plan_internal.h"

namespace mfm {

PIdx pidx_for(int variant) {
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

int build(MfmPlan* P) {
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
    // default from T*B = 2560 rows (B = 128 at T = 20: where the bf16 recurrences start since round 5; at B = 144 the bf16
    // recurrences WITHOUT bf16-resident buffers are 0.3193 ms, with them 0.2786).  Rounds 2-4: 3840.  Measured at the MOSI sizes
    // (bf16-resident vs fp32-stored, ms per step; B = 192 / 256 / 384 / 512 / 768 / 1024): 0.366 vs 0.377, 0.377 vs 0.418,
    // 0.406 vs 0.471, 0.418 vs 0.521, 0.450 vs 0.603, 0.481 vs 0.706 (round 2: crossover at T*B = 16384; then proj_bf16.hip, the
    // 64-row decoder fc1 and whole rounds of workgroups in the one-pass weight-gradient launch); MFM_BF16_STORE=1 forces it
    // on for every size, =0 off
    const char* se = opt_get("MFM_BF16_STORE");
    long st_minrows = 2560;
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
  L.tail_from = st;        // f_l, f_a, f_v, f_y are final: everything behind this stage feeds the losses only
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
    bool ok = c.B <= lat_row_maxb && !split0 && (size_t)2 * rs * sizeof(float) <= 24 * 1024 && P->n_params < (1ll << 29);      // (byte offsets of the buffer loads: 32 bits)
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
    // bias gradients (round 6): thread t of a chain workgroup adds ONE element -- x = its offset in the gradient buffer (-1: none),
    // y = its place in the gradient record -- tabulated in the LAST stage slot of the backward table (free while nstages < 8)
    // instead of a search through the op table per element at the end of the backward chain
    L.n_params = P->n_params;
    L.bias_tab = 0;
    if (L.nstages < MFM_LAT_MAXSTAGES) {
      bool fits = true;
      for (int ch = 0; ch < L.nch && fits; ++ch) {
        int cnt = 0;
        for (int t = 0; t < NT; ++t) { int* e = bw + (((size_t)ch * MFM_LAT_MAXSTAGES + MFM_LAT_MAXSTAGES - 1) * NT + t) * 4; e[0] = -1; e[1] = e[2] = e[3] = 0; }
        for (int i = 0; i < L.nops && fits; ++i) {
          const LatOp& op = P->lat_ops[i];
          if (L.nch > 1 && op.chain != ch) continue;
          for (int n = 0; n < op.N; ++n) {
            if (cnt >= NT || op.b_off + n >= (1ll << 31)) { fits = false; break; }
            int* e = bw + (((size_t)ch * MFM_LAT_MAXSTAGES + MFM_LAT_MAXSTAGES - 1) * NT + cnt) * 4;
            e[0] = (int)(op.b_off + n); e[1] = op.out_off + n;
            ++cnt;
          }
        }
      }
      L.bias_tab = fits ? 1 : 0;
    }
    L.bias_n = 0;
    for (int ch = 0; ch < L.nch && L.bias_tab; ++ch) {
      int cnt = 0;
      for (int i = 0; i < L.nops; ++i) if (L.nch == 1 || P->lat_ops[i].chain == ch) cnt += P->lat_ops[i].N;
      L.bias_n = std::max(L.bias_n, cnt);
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
      if (i >= P->n_enc && (hh & 3) == 0) { P->wf_img[i - P->n_enc] = carve(cur, 4 * HKB * HKB); P->wf_img[3 + i - P->n_enc] = carve(cur, 4 * HKB * HKB); }
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

}  // namespace mfm
