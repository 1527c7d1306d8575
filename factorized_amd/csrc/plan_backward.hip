// The fused plan, backward chain B0..B5 (reference: loss.backward() of train_mfm / train_beta_vae, mfm_mosi.py:440 / 282): decoder
// fc1 -> decoder BPTT -> latent stack -> (MFN) -> encoder BPTT -> every weight gradient; at B <= 32 the last three share ONE
// launch (role workgroups, dw_role_dev.h) whose block table is built here.
#include "plan_internal.h"

namespace mfm {

void dA_gemms(const MfmPlan* P, const SeqBuf& sb, int pb, float* W, float* grads, std::vector<MfmGemmDesc>& out,
                     const float* xin, int64_t ldx, int kin, bool dec, bool only_init) {
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

int dw_role_build(MfmPlan* P, const std::vector<MfmGemmDesc>& all, float* W, int key, hipStream_t s, DwRole* out) {
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


int backward(MfmPlan* P, const float* params, const float* x, const void* y, int stage, float* W,
                    float* grads, hipStream_t s, const ExtGrads* ext, const LossW* lw) {
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
        MFM_LAUNCH_TIMED(guard_propagate_kernel, dim3(1), dim3(64), 0, s, P->status_ptr(W), guard);
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
  LatentDev Lbase = P->lat;
  {
    LatentDev& L = Lbase;
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
  }
  bool bwd_head_done = false;
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
      // MFM_KL_EF at small batches: the part of the latent backward chains that does not wait for the decoders -- seeds of the
      // discriminative and KLD terms, classifier and logvar stages -- runs on this launch's idle CUs (head blocks); the chain in
      // front of the encoder BPTT then starts from their record (LatentDev::bwd_split)
      bool dec_bwd_done = false;
      if (imgs_on && V == 0 && P->n_enc == 4 && P->fold_state == 1 && !Lbase.pre && Lbase.row_path) {
        LatentDev LH = Lbase;
        LH.gen_w = 0.0f;
        for (int m = 0; m < 3; ++m) LH.d_dec_init[m] = nullptr;
        int rc;
        { Timer _t(P, s, K_DEC_BWD); rc = seq_dec_bwd_head_launch(q, 3, T, B, dimg, LH, params, grads, s); }
        if (rc == MFM_OK) { dec_bwd_done = true; bwd_head_done = true; }
        else if (rc != MFM_ERR_UNSUPPORTED) return rc;
      }
      if (dec_bwd_done) {}
      else if (imgs_on) RUN(K_DEC_BWD, seq_bwd_img_launch(q, 3, T, B, dimg, s));
      else RUN(K_DEC_BWD, seq_bf16 ? mfm_lstm_seq_bwd_bf16(q, 3, T, B, s) : mfm_lstm_seq_bwd(q, 3, T, B, s));
    }
    // (B2: the decoder weight gradients only feed Adam; they share the encoders' launch at the end)
  }
  // B3: latent stack
  bool enc_bwd_done = false;
  {
    LatentDev L = Lbase;
    L.bwd_split = bwd_head_done ? 1 : 0;
    if (bwd_head_done) { L.grd_seed = W + P->lat_grd; L.seed_w = 1.0f; L.seed_w_ptr = nullptr; }      // the head blocks' record
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
          !opt_get("MFM_DW_F32_MINROWS") && !(opt_get("MFM_GEMM_TN") && atoi(opt_get("MFM_GEMM_TN")) == 0)) {
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
              MFM_LAUNCH_TIMED(tick_kernel, dim3(1), dim3(64), 0, s, (unsigned long long*)nullptr, P->dw_tick_ptr(W));
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
  // B5: all weight gradients on the grouped TN GEMM (a one-pass kernel over dA was built in round 2, measured slower at
  // B = 2048 and removed in round 5: profiles/r02_dw_onepass.txt)
  {
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
    const bool f32pass = !c.precision && f32_min_rows > 0 && TB >= f32_min_rows && TB > 1;
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
    for (int e = 0; e < P->n_enc && !st16; ++e) {
      if (f32_done[e]) continue;
      dA_gemms(P, P->enc[e], P->enc_p[e], W, grads, tail, x + P->enc_xoff[e], P->D, P->enc_d[e], false);
    }
    if (gen_on && !st16)
      for (int m = 0; m < 3; ++m)
        dA_gemms(P, P->dec[m], P->dec_p[m], W, grads, tail, W + P->dec_init[m], P->dec_h[m], P->dec_h[m], true, f32_done[6 + m]);
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
