// The fused plan, Memory Fusion Network part (variants MFM_KL / MFM; reference mfm_model.py:93-199): the launches between the
// three MFN LSTMs and the latent stack, forward and backward.
#include "plan_internal.h"

namespace mfm {

// ---- Memory Fusion Network, forward (reference mfm_model.py:140-199 restructured, see mfn_att.hip / mfn_mem.hip):
// cStar gather -> att1_fc1 (+relu/dropout) -> att1_fc2 -> softmax * cStar -> {att2_fc1 (+relu/dropout), attended part
// of gamma1_fc1 / gamma2_fc1} -> att2_fc2 (+tanh) -> memory recurrence -> heads on [h_l, h_a, h_v, mem]
bool mfn_heads_desc(const MfmPlan* P, const float* params, float* W, MfnHeadsDev& H) {
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

int mfn_forward(MfmPlan* P, const float* params, int train, uint64_t seed, float* W, hipStream_t s) {
  const MfmPlanConfig& c = P->cfg;
  const PIdx& pi = P->pi;
  const int T = P->T, B = P->B, M = c.mem_dim, A2 = P->A2;
  const int64_t TB = (int64_t)T * B;
  const int prec = c.precision;
  GemmEpiSet es;
  memset(&es, 0, sizeof(es));
  es.seed = seed * 0x9E3779B97F4A7C15ull + P->calls * 0xD1B54A32D192ED03ull; es.train = train;
  es.tick = reinterpret_cast<const unsigned long long*>(P->tick_ptr(W));
  {
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


// ---- Memory Fusion Network, backward.  Appends the MFN's weight-gradient products to `tail`.
int mfn_backward(MfmPlan* P, const float* params, float* W, float* grads, hipStream_t s,
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
  {
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

}  // namespace mfm
