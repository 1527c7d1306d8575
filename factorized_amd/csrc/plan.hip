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
#include "lstm_seq_dev.h"

namespace mfm {

enum KernelId {
  K_PROJ = 0, K_ENC_FWD, K_LAT_FWD, K_DEC_FWD, K_FC1_FWD, K_MSE, K_FC1_BWD, K_DEC_BWD, K_DEC_DW,
  K_LAT_BWD, K_ENC_BWD, K_ENC_DW, K_ADAM, K_LAT_DW, K_PACK, K_COUNT
};

// state_dict order of MFM_KL_EF (78 tensors), see include/mfm_hip.h
enum { P_ENC_L = 0, P_ENC_A = 6, P_ENC_V = 12, P_DEC_L = 18, P_DEC_A = 24, P_DEC_V = 30, P_ENC_Y = 36,
       P_TO_ZY = 42, P_TO_LVY = 44, P_TO_ZL = 46, P_TO_ZA = 48, P_TO_ZV = 50, P_TO_LVL = 52, P_TO_LVA = 54,
       P_TO_LVV = 56, P_ZY_F1 = 58, P_ZY_F2 = 60, P_ZL_F1 = 62, P_ZL_F2 = 64, P_ZA_F1 = 66, P_ZA_F2 = 68,
       P_ZV_F1 = 70, P_ZV_F2 = 72, P_Y_F1 = 74, P_Y_F2 = 76 };
enum { W_IH = 0, W_HH = 1, B_IH = 2, B_HH = 3, FC_W = 4, FC_B = 5 };

struct SeqBuf { int64_t gates, hs, cs, wpack; int h, Hp; };

struct TimingPair { hipEvent_t a, b; int kid; };

}  // namespace mfm

struct MfmPlan {
  MfmPlanConfig cfg;
  int64_t off[MFM_KLEF_NPARAM];
  int64_t n_params;
  int D, T, B;
  // encoders l,a,v,y ; decoders l,a,v
  int enc_d[4], enc_xoff[4], enc_h[4], enc_p[4];
  int dec_d[3], dec_h[3], dec_p[3], dec_xoff[3];
  mfm::SeqBuf enc[4], dec[3];
  int64_t dec_dhs[3], dec_init[3], dec_dinit[3], xhat[3], dxhat[3];
  int64_t lat_rec, dh_last[4], yhat, ones, losses;
  int64_t ws_floats;
  mfm::LatentDev lat;
  mfm::LatOp lat_ops[MFM_LAT_MAXOPS];
  int64_t lat_ops_off, dbg_off, lat_grd, lat_items_off;
  int lay_f1[4], lay_m1[4], lay_c1, lay_mc;   // record offsets kept for mfm_plan_latent_layout
  std::vector<int> lat_items;       // row-path item tables: forward then backward, [MAXSTAGES][1024][4] each
  // timing
  int timing_mask;
  std::vector<mfm::TimingPair> pool;
  size_t pool_used;
  uint64_t calls;
  const float* grads_prezeroed;     // gradient buffer cleared by the forward pass of the running fused step
};

namespace mfm {

static int64_t carve(int64_t& cursor, int64_t n) {
  const int64_t at = cursor;
  cursor = round_up64(cursor + n, 64);   // 256-byte granules
  return at;
}

static void add_op(LatOp* ops, LatentDev& L, int stage, int in_off, int out_off, int K, int N, int64_t w_off, int64_t b_off,
                   int relu, int mask_off, float p) {
  LatOp& o = ops[L.nops++];
  o.in_off = in_off; o.out_off = out_off; o.K = K; o.N = N; o.w_off = w_off; o.b_off = b_off;
  o.relu = relu; o.mask_off = mask_off; o.drop_p = p; o.stage = stage;
}

static int build(MfmPlan* P) {
  const MfmPlanConfig& c = P->cfg;
  P->T = c.T; P->B = c.B;
  P->D = c.d_l + c.d_a + c.d_v;
  const int ze = c.zl + c.za + c.zv;
  const int ed[4] = {c.d_l, c.d_a, c.d_v, P->D};
  const int ex[4] = {0, c.d_l, c.d_l + c.d_a, 0};
  const int eh[4] = {c.zl, c.za, c.zv, ze};
  const int ep[4] = {P_ENC_L, P_ENC_A, P_ENC_V, P_ENC_Y};
  const int dd[3] = {c.d_l, c.d_a, c.d_v};
  const int fm[3] = {c.fl, c.fa, c.fv};
  const int dp[3] = {P_DEC_L, P_DEC_A, P_DEC_V};
  int64_t cur = 0;
  const int64_t TB = (int64_t)c.T * c.B;
  for (int e = 0; e < 4; ++e) {
    P->enc_d[e] = ed[e]; P->enc_xoff[e] = ex[e]; P->enc_h[e] = eh[e]; P->enc_p[e] = ep[e];
    SeqBuf& s = P->enc[e];
    s.h = eh[e]; s.Hp = round_up(eh[e], 16);
    s.gates = carve(cur, TB * 4 * s.Hp);
    s.hs = carve(cur, TB * s.Hp);
    s.cs = carve(cur, TB * s.Hp);
    s.wpack = c.precision ? carve(cur, mfm_lstm_pack_bytes(s.h, 0) / 4) : -1;
    P->dh_last[e] = carve(cur, (int64_t)c.B * eh[e]);
  }
  for (int m = 0; m < 3; ++m) {
    P->dec_d[m] = dd[m]; P->dec_h[m] = c.fy + fm[m]; P->dec_p[m] = dp[m]; P->dec_xoff[m] = ex[m];
    SeqBuf& s = P->dec[m];
    s.h = P->dec_h[m]; s.Hp = round_up(s.h, 16);
    s.gates = carve(cur, TB * 4 * s.Hp);
    s.hs = carve(cur, TB * s.Hp);
    s.cs = carve(cur, TB * s.Hp);
    s.wpack = c.precision ? carve(cur, mfm_lstm_pack_bytes(s.h, 1) / 4) : -1;
    P->dec_dhs[m] = carve(cur, TB * s.Hp);
    P->dec_init[m] = carve(cur, (int64_t)c.B * s.h);
    P->dec_dinit[m] = carve(cur, (int64_t)c.B * s.h);
    P->xhat[m] = carve(cur, TB * dd[m]);
    P->dxhat[m] = carve(cur, TB * dd[m]);
  }
  // ---- latent record layout (every segment starts on a multiple of 4 floats)
  LatentDev& L = P->lat;
  memset(&L, 0, sizeof(L));
  int rs = 0;
  auto seg = [&](int n) { const int at = rs; rs += round_up(n, 4); return at; };
  const int zn[4] = {c.zl, c.za, c.zv, c.zy};
  const int fn[4] = {c.fl, c.fa, c.fv, c.fy};
  int last_off[4], f1_off[4], m1_off[4];
  for (int e = 0; e < 4; ++e) { L.in_off[e] = seg(eh[e]); L.enc_n[e] = eh[e]; }
  for (int e = 0; e < 4; ++e) last_off[e] = seg(eh[e]);
  for (int e = 0; e < 4; ++e) { L.mu_off[e] = seg(zn[e]); L.z_n[e] = zn[e]; }
  for (int e = 0; e < 4; ++e) L.lv_off[e] = seg(zn[e]);
  for (int e = 0; e < 4; ++e) { f1_off[e] = seg(fn[e]); m1_off[e] = seg(fn[e]); }
  for (int e = 0; e < 4; ++e) { L.f_off[e] = seg(fn[e]); L.f_n[e] = fn[e]; }
  const int c1_off = seg(c.fy), mc_off = seg(c.fy);
  L.yhat_off = seg(c.output_dim); L.od = c.output_dim;
  L.rec_size = rs;
  for (int e = 0; e < 4; ++e) { P->lay_f1[e] = f1_off[e]; P->lay_m1[e] = m1_off[e]; }
  P->lay_c1 = c1_off; P->lay_mc = mc_off;
  const int64_t* o = P->off;
  // stage 0: encoder fc1 (mfm_model.py:60-61)
  for (int e = 0; e < 4; ++e)
    add_op(P->lat_ops, L, 0, L.in_off[e], last_off[e], eh[e], eh[e], o[ep[e] + FC_W], o[ep[e] + FC_B], 0, -1, 0.f);
  // stage 1: mu heads (mfm_model.py:630-639).  The logvar heads only feed the KLD, nothing downstream waits
  // for them, so they ride along with the classifier's first layer in stage 4 (the row kernels give every
  // thread one work item per stage: 4*(16+152) output quads and 4*(16+240)/4 input groups still fit 1024).
  const int pmu[4] = {P_TO_ZL, P_TO_ZA, P_TO_ZV, P_TO_ZY};
  const int plv[4] = {P_TO_LVL, P_TO_LVA, P_TO_LVV, P_TO_LVY};
  for (int e = 0; e < 4; ++e)
    add_op(P->lat_ops, L, 1, last_off[e], L.mu_off[e], eh[e], zn[e], o[pmu[e]], o[pmu[e] + 1], 0, -1, 0.f);
  // stages 2, 3: z -> f MLPs (mfm_model.py:644-647)
  const int pf1[4] = {P_ZL_F1, P_ZA_F1, P_ZV_F1, P_ZY_F1};
  const int pf2[4] = {P_ZL_F2, P_ZA_F2, P_ZV_F2, P_ZY_F2};
  const float pd[4] = {c.drop_zl, c.drop_za, c.drop_zv, c.drop_zy};
  for (int e = 0; e < 4; ++e)
    add_op(P->lat_ops, L, 2, L.mu_off[e], f1_off[e], zn[e], fn[e], o[pf1[e]], o[pf1[e] + 1], 1, m1_off[e], pd[e]);
  for (int e = 0; e < 4; ++e)
    add_op(P->lat_ops, L, 3, f1_off[e], L.f_off[e], fn[e], fn[e], o[pf2[e]], o[pf2[e] + 1], 1, -1, 0.f);
  // stages 4, 5: classifier (mfm_model.py:657); stage 4 also carries the logvar heads
  add_op(P->lat_ops, L, 4, L.f_off[3], c1_off, c.fy, c.fy, o[P_Y_F1], o[P_Y_F1 + 1], 1, mc_off, c.drop_y);
  for (int e = 0; e < 4; ++e)
    add_op(P->lat_ops, L, 4, last_off[e], L.lv_off[e], eh[e], zn[e], o[plv[e]], o[plv[e] + 1], 0, -1, 0.f);
  add_op(P->lat_ops, L, 5, c1_off, L.yhat_off, c.fy, c.output_dim, o[P_Y_F2], o[P_Y_F2 + 1], 0, -1, 0.f);
  L.nstages = 6;
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
  L.has_logvar = 1;
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
  if (((size_t)panel + 2 * (size_t)rs) * sizeof(float) <= LDS_BUDGET) {
    L.wpanel = panel;
    while (R > 1 && (2 * (size_t)R * rs + panel) * sizeof(float) > LDS_BUDGET) R >>= 1;
  } else {
    L.wpanel = 0;   // stage tensors not contiguous / too large to stage: kernels read them from L2
    while (R > 1 && 2 * (size_t)R * rs * sizeof(float) > LDS_BUDGET) R >>= 1;
  }
  L.rows_per_wg = R;
  // Latency path (latent.hip, row kernels): one row per workgroup while that still fits the chip in one
  // wave of workgroups and every layer meets the vector-load shape requirements.
  {
    const int row_maxb = getenv("MFM_LATENT_ROW_MAXB") ? atoi(getenv("MFM_LATENT_ROW_MAXB")) : 256;   // tuning override
    bool ok = c.B <= row_maxb && (size_t)2 * rs * sizeof(float) <= 24 * 1024 && P->n_params < (1ll << 31);
    for (int i = 0; i < L.nops && ok; ++i) {
      const LatOp& op = P->lat_ops[i];
      ok = (op.K % 4 == 0) && op.K >= 4 && op.K <= 128 && op.N <= 128 && (op.w_off % 4 == 0);
    }
    for (int st = 0; st < L.nstages && ok; ++st) {
      int sn = 0, sk = 0;
      for (int i = L.stage_begin[st]; i < L.stage_begin[st + 1]; ++i) { sn += P->lat_ops[i].N; sk += P->lat_ops[i].K; }
      ok = 4 * sn <= 1024 && 4 * sk <= 1024;      // one work item per thread and stage
    }
    if (const char* e = getenv("MFM_LATENT_PATH")) { if (!strcmp(e, "staged")) ok = false; }
    L.row_path = ok ? 1 : 0;
  }
  // row path: the work item of thread t in stage s is static, so it is tabulated here once (encoding: latent.hip)
  const int NT = MFM_LAT_ROW_THREADS;
  P->lat_items.assign((size_t)2 * MFM_LAT_MAXSTAGES * NT * 4, 0);
  if (L.row_path) {
    int* fw = P->lat_items.data();
    int* bw = fw + (size_t)MFM_LAT_MAXSTAGES * NT * 4;
    for (int st = 0; st < L.nstages; ++st) {
      const int ob = L.stage_begin[st], oe = L.stage_begin[st + 1];
      int sn = 0, sk = 0;
      for (int i = ob; i < oe; ++i) { sn += P->lat_ops[i].N; sk += P->lat_ops[i].K; }
      L.nitems_fwd[st] = 4 * sn;
      L.nitems_bwd[st] = 4 * sk;
      for (int t = 0; t < NT; ++t) {
        {   // forward: quad (n, q) -> output column n of op o
          const bool live = t < 4 * sn;
          const int item = std::min(t, 4 * sn - 1) >> 2;
          int o = ob;
          while (o + 1 < oe && item >= P->lat_ops[o + 1].pfx_n) ++o;
          const LatOp& op = P->lat_ops[o];
          const int n = item - op.pfx_n;
          int* e = fw + ((size_t)st * NT + t) * 4;
          e[0] = (int)(op.w_off + (int64_t)n * op.K);
          e[1] = (int)(op.b_off + n);
          e[2] = op.in_off | (op.K << 16);
          e[3] = (op.out_off + n) | (o << 16) | ((op.relu ? 1 : 0) << 24) | ((op.mask_off >= 0 ? 1 : 0) << 25) |
                 ((live ? 1 : 0) << 26);
        }
        {   // backward: 16 lanes (kc, l) -> input columns kc..kc+3 of op o
          const bool live = t < 4 * sk;
          const int col = (std::min(t, 4 * sk - 1) >> 4) * 4;
          int o = ob;
          while (o + 1 < oe && col >= P->lat_ops[o + 1].pfx_k) ++o;
          const LatOp& op = P->lat_ops[o];
          const int kc = col - op.pfx_k;
          int* e = bw + ((size_t)st * NT + t) * 4;
          e[0] = (int)(op.w_off + kc);
          e[1] = op.K | (op.N << 8);
          e[2] = op.out_off | ((op.in_off + kc) << 16);
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
          e[3] = (live ? 1 : 0) | (prelu << 1) | (pmask << 2);
        }
      }
    }
  }

  P->lat_ops_off = carve(cur, (int64_t)(sizeof(P->lat_ops) / (sizeof(float))));
  P->dbg_off = carve(cur, 128);     // 64 x u64 debug timestamps
  P->lat_items_off = carve(cur, (int64_t)P->lat_items.size());
  P->lat_grd = carve(cur, (int64_t)c.B * rs);
  P->lat_rec = carve(cur, (int64_t)c.B * rs);
  P->yhat = carve(cur, (int64_t)c.B * c.output_dim);
  P->ones = carve(cur, TB);
  P->losses = carve(cur, MFM_LOSS_SLOTS);
  P->ws_floats = cur;
  return MFM_OK;
}

struct Timer {
  MfmPlan* P; hipStream_t s; int kid; TimingPair* tp;
  Timer(MfmPlan* P_, hipStream_t s_, int kid_) : P(P_), s(s_), kid(kid_), tp(nullptr) {
    if (!(P->timing_mask & (1 << kid))) return;
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
  return d;
}

static int forward(MfmPlan* P, const float* params, const float* x, const void* y, int train, uint64_t seed,
                   float* W, float* xhat_out[3], float* yhat_out, float* losses_out, hipStream_t s,
                   float* grads_to_zero = nullptr) {
  const MfmPlanConfig& c = P->cfg;
  const int T = P->T, B = P->B;
  const int64_t TB = (int64_t)T * B;
  float* losses = losses_out ? losses_out : W + P->losses;
  // the loss slots (and, in the fused step, the gradient buffer) are cleared by the first GEMM launch
  ZeroSpans zs;
  memset(&zs, 0, sizeof(zs));
  if ((((uintptr_t)losses) & 15) == 0) { zs.ptr[0] = losses; zs.n[0] = MFM_LOSS_SLOTS; }
  else MFM_HIP_CHECK(hipMemsetAsync(losses, 0, MFM_LOSS_SLOTS * sizeof(float), s));
  P->grads_prezeroed = nullptr;
  if (grads_to_zero && (((uintptr_t)grads_to_zero) & 15) == 0 && (P->n_params & 3) == 0) {
    zs.ptr[1] = grads_to_zero; zs.n[1] = P->n_params;
    P->grads_prezeroed = grads_to_zero;
  }
  P->calls++;

  // bf16 plans: the recurrences' weight fragments, rounded and packed once per step (lstm_seq_bf16.hip)
  const bool seq_bf16 = c.precision && bf16_seq_pays(B);
  if (seq_bf16) {
    MfmSeqDesc q[7];
    for (int e = 0; e < 4; ++e) q[e] = seq_desc(P, P->enc[e], P->enc_p[e], params, W, false);
    for (int m = 0; m < 3; ++m) q[4 + m] = seq_desc(P, P->dec[m], P->dec_p[m], params, W, true);
    RUN(K_PACK, mfm_lstm_pack_bf16(q, 7, s));
  }

  // F0: input projections
  {
    MfmGemmDesc g[4];
    memset(g, 0, sizeof(g));
    for (int e = 0; e < 4; ++e) {
      const SeqBuf& sb = P->enc[e];
      const int pb = P->enc_p[e];
      MfmGemmDesc& d = g[e];
      d.a = x + P->enc_xoff[e]; d.a_sm = P->D; d.a_sk = 1; d.a_sz = 0;
      d.b = params + P->off[pb + W_IH]; d.b_sz = (int64_t)sb.h * P->enc_d[e]; d.b_sn = P->enc_d[e]; d.b_sk = 1;
      d.c = W + sb.gates; d.c_sz = sb.Hp; d.ldc = 4 * (int64_t)sb.Hp;
      d.bias = params + P->off[pb + B_IH]; d.bias2 = params + P->off[pb + B_HH]; d.bias_sz = sb.h;
      d.m = (int)TB; d.n = sb.Hp; d.n_valid = sb.h; d.k = P->enc_d[e]; d.batch = 4; d.split_k = 1;
      d.alpha = 1.0f;
    }
    RUN(K_PROJ, gemm_group_launch(g, 4, s, &zs, nullptr, 0, c.precision));
  }
  // F1: encoder recurrences
  {
    MfmSeqDesc q[4];
    for (int e = 0; e < 4; ++e) q[e] = seq_desc(P, P->enc[e], P->enc_p[e], params, W, false);
    RUN(K_ENC_FWD, seq_bf16 ? mfm_lstm_seq_fwd_bf16(q, 4, T, B, s) : mfm_lstm_seq_fwd(q, 4, T, B, s));
  }
  // F2: latent stack
  {
    LatentDev L = P->lat;
    L.ops = reinterpret_cast<const LatOp*>(W + P->lat_ops_off);
    L.items_fwd = reinterpret_cast<const int*>(W + P->lat_items_off);
    L.items_bwd = L.items_fwd + (size_t)MFM_LAT_MAXSTAGES * MFM_LAT_ROW_THREADS * 4;
    if (getenv("MFM_LATENT_DBG")) L.dbg = reinterpret_cast<unsigned long long*>(W + P->dbg_off);
    for (int e = 0; e < 4; ++e) {
      L.enc_h[e] = W + P->enc[e].hs + (int64_t)(T - 1) * B * P->enc[e].Hp;
      L.enc_ld[e] = P->enc[e].Hp;
    }
    for (int m = 0; m < 3; ++m) { L.dec_init[m] = W + P->dec_init[m]; L.dec_ld[m] = P->dec_h[m]; }
    L.rec = W + P->lat_rec;
    L.yhat_out = yhat_out ? yhat_out : W + P->yhat;
    L.y = y; L.losses = losses; L.train = train;
    L.seed = seed * 0x9E3779B97F4A7C15ull + P->calls;
    RUN(K_LAT_FWD, latent_fwd_launch(L, params, s));
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
      d.a = W + sb.hs; d.a_sm = sb.Hp; d.a_sk = 1;
      d.b = params + P->off[pb + FC_W]; d.b_sn = sb.h; d.b_sk = 1;
      d.c = xh[m]; d.ldc = P->dec_d[m];
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
    }
    RUN(K_FC1_FWD, gemm_group_launch(g, 3, s, nullptr, me, 3, c.precision));
  }
  return MFM_OK;
}

static void dA_gemms(const MfmPlan* P, const SeqBuf& sb, int pb, float* W, float* grads, std::vector<MfmGemmDesc>& out,
                     const float* xin, int64_t ldx, int kin, bool dec) {
  const int T = P->T, B = P->B;
  const int64_t TB = (int64_t)T * B;
  MfmGemmDesc base;
  memset(&base, 0, sizeof(base));
  base.a_sz = sb.Hp; base.a_sm = 1; base.a_sk = 4 * (int64_t)sb.Hp;
  base.m = sb.h; base.batch = 4; base.accumulate = 1; base.split_k = 0; base.alpha = 1.0f;
  // recurrent product sum_{t>=1} dA_t^T h_{t-1}
  if (T > 1) {
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
  {
    MfmGemmDesc d = base;
    d.a = W + sb.gates;
    d.b = W + P->ones; d.b_sk = 1; d.b_sn = 1;
    d.k = (int)TB; d.n = 1; d.n_valid = 1;
    d.c = grads + P->off[pb + B_IH]; d.c_sz = sb.h; d.ldc = 1;
    d.c2 = grads + P->off[pb + B_HH];
    out.push_back(d);
  }
}

struct ExtGrads {           // upstream gradients supplied by the caller (autograd module path)
  const float* d_xhat[3];
  const float* d_yhat;
  const float* d_reg;       // device scalar
};

static int backward(MfmPlan* P, const float* params, const float* x, const void* y, int stage, float* W,
                    float* grads, hipStream_t s, const ExtGrads* ext = nullptr) {
  const MfmPlanConfig& c = P->cfg;
  const int T = P->T, B = P->B;
  const int64_t TB = (int64_t)T * B;
  if (P->grads_prezeroed != grads) MFM_HIP_CHECK(hipMemsetAsync(grads, 0, (size_t)P->n_params * sizeof(float), s));
  P->grads_prezeroed = nullptr;
  const bool gen_on = (stage != 2), disc_on = (stage != 1);
  const bool seq_bf16 = c.precision && bf16_seq_pays(B);
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
      d.a = dxh; d.a_sm = P->dec_d[m]; d.a_sk = 1;
      d.b = params + P->off[pb + FC_W]; d.b_sk = sb.h; d.b_sn = 1;
      d.c = W + P->dec_dhs[m]; d.ldc = sb.Hp;
      d.m = (int)TB; d.n = sb.Hp; d.n_valid = sb.h; d.k = P->dec_d[m]; d.split_k = 1;
      g.push_back(d);
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
    RUN(K_FC1_BWD, gemm_group_launch(g.data(), (int)g.size(), s, nullptr, nullptr, 0, c.precision));
    // B1: decoder BPTT
    {
      MfmSeqDesc q[3];
      for (int m = 0; m < 3; ++m) {
        q[m] = seq_desc(P, P->dec[m], P->dec_p[m], params, W, true);
        q[m].h_init = W + P->dec_init[m]; q[m].ld_init = P->dec_h[m];
        q[m].dh_ext = W + P->dec_dhs[m]; q[m].ld_dh = P->dec[m].Hp;
        q[m].d_h_init = W + P->dec_dinit[m]; q[m].ld_dinit = P->dec_h[m];
      }
      RUN(K_DEC_BWD, seq_bf16 ? mfm_lstm_seq_bwd_bf16(q, 3, T, B, s) : mfm_lstm_seq_bwd(q, 3, T, B, s));
    }
    // (B2: the decoder weight gradients only feed Adam; they share the encoders' launch at the end)
  }
  // B3: latent stack
  {
    LatentDev L = P->lat;
    L.ops = reinterpret_cast<const LatOp*>(W + P->lat_ops_off);
    L.items_fwd = reinterpret_cast<const int*>(W + P->lat_items_off);
    L.items_bwd = L.items_fwd + (size_t)MFM_LAT_MAXSTAGES * MFM_LAT_ROW_THREADS * 4;
    if (getenv("MFM_LATENT_DBG")) L.dbg = reinterpret_cast<unsigned long long*>(W + P->dbg_off);
    for (int m = 0; m < 3; ++m) {
      L.d_dec_init[m] = gen_on ? W + P->dec_dinit[m] : nullptr;
      L.dec_ld[m] = P->dec_h[m];
    }
    for (int e = 0; e < 4; ++e) { L.dh_last[e] = W + P->dh_last[e]; L.dh_ld[e] = P->enc_h[e]; }
    L.rec = W + P->lat_rec;
    L.y = y;
    L.grd_out = W + P->lat_grd;
    if (ext) { L.d_yhat_ext = ext->d_yhat; L.reg_w_ptr = ext->d_reg; }
    L.reg_w = c.lda_reg * c.reg_scale;
    L.disc_w = disc_on ? 1.0f : 0.0f;
    L.gen_w = gen_on ? 1.0f : 0.0f;
    RUN(K_LAT_BWD, latent_bwd_launch(L, params, grads, s));
    // weight gradients of the 22 latent Linears: dW[n][k] = sum_r G[r][out+n] X[r][in+k]
    const int rs = P->lat.rec_size;
    for (int i = 0; i < P->lat.nops; ++i) {
      const LatOp& op = P->lat_ops[i];
      MfmGemmDesc d;
      memset(&d, 0, sizeof(d));
      d.alpha = 1.0f; d.batch = 1; d.split_k = 1; d.accumulate = 0;
      d.a = W + P->lat_grd + op.out_off; d.a_sm = 1; d.a_sk = rs;
      d.b = W + P->lat_rec + op.in_off; d.b_sk = rs; d.b_sn = 1;
      d.c = grads + op.w_off; d.ldc = op.K;
      d.m = op.N; d.n = op.K; d.n_valid = op.K; d.k = P->B;
      tail.push_back(d);
    }
  }
  // B4: encoder BPTT
  {
    MfmSeqDesc q[4];
    for (int e = 0; e < 4; ++e) {
      q[e] = seq_desc(P, P->enc[e], P->enc_p[e], params, W, false);
      q[e].dh_ext = W + P->dh_last[e]; q[e].ld_dh = P->enc_h[e];
    }
    RUN(K_ENC_BWD, seq_bf16 ? mfm_lstm_seq_bwd_bf16(q, 4, T, B, s) : mfm_lstm_seq_bwd(q, 4, T, B, s));
  }
  // B5: all weight gradients
  {
    for (int e = 0; e < 4; ++e)
      dA_gemms(P, P->enc[e], P->enc_p[e], W, grads, tail, x + P->enc_xoff[e], P->D, P->enc_d[e], false);
    if (gen_on)
      for (int m = 0; m < 3; ++m)
        dA_gemms(P, P->dec[m], P->dec_p[m], W, grads, tail, W + P->dec_init[m], P->dec_h[m], P->dec_h[m], true);
    RUN(K_ENC_DW, c.precision ? mfm_gemm_grouped_bf16(tail.data(), (int)tail.size(), s)
                                : mfm_gemm_grouped_f32(tail.data(), (int)tail.size(), s));
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
  MfmPlan* P = new (std::nothrow) MfmPlan();
  if (!P) { set_error("plan: out of host memory"); return MFM_ERR_ARG; }
  P->cfg = c;
  if (P->cfg.reg_scale == 0.0f) P->cfg.reg_scale = 1.0f;
  for (int i = 0; i < MFM_KLEF_NPARAM; ++i) {
    P->off[i] = param_offsets[i];
    if (param_offsets[i] < 0 || param_offsets[i] >= n_params_total) {
      delete P; set_error("plan: param offset %d out of range", i); return MFM_ERR_ARG;
    }
  }
  P->n_params = n_params_total;
  P->timing_mask = 0; P->pool_used = 0; P->calls = 0; P->grads_prezeroed = nullptr;
  int rc = build(P);
  if (rc != MFM_OK) { delete P; return rc; }
  *out = P;
  return MFM_OK;
}

extern "C" void mfm_plan_destroy(MfmPlan* P) {
  if (!P) return;
  for (auto& t : P->pool) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
  delete P;
}

extern "C" int64_t mfm_plan_debug_offset(const MfmPlan* P) { return P ? P->dbg_off * (int64_t)sizeof(float) : -1; }

extern "C" int64_t mfm_plan_workspace_bytes(const MfmPlan* P) { return P ? P->ws_floats * (int64_t)sizeof(float) : 0; }

extern "C" int mfm_plan_init_workspace(MfmPlan* P, void* workspace, void* stream) {
  if (!P || !workspace) { set_error("mfm_plan_init_workspace: null argument"); return MFM_ERR_ARG; }
  float* W = (float*)workspace;
  hipStream_t s = (hipStream_t)stream;
  MFM_HIP_CHECK(hipMemsetAsync(W, 0, (size_t)P->ws_floats * sizeof(float), s));
  MFM_HIP_CHECK(hipMemcpyAsync(W + P->lat_ops_off, P->lat_ops, sizeof(P->lat_ops), hipMemcpyHostToDevice, s));
  MFM_HIP_CHECK(hipMemcpyAsync(W + P->lat_items_off, P->lat_items.data(), P->lat_items.size() * sizeof(int),
                               hipMemcpyHostToDevice, s));
  return fill_launch(W + P->ones, (int64_t)P->T * P->B, 1.0f, s);
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
  rc = backward(P, params, x, y, 0, (float*)workspace, grads, s);
  if (rc != MFM_OK) return rc;
  RUN(K_ADAM, adam_launch(params, grads, adam_m, adam_v, P->n_params, step, lr, 0.9f, 0.999f, 1e-8f, grad_scale, s));
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
  RUN(K_ADAM, adam_spans_launch(params, grads, adam_m, adam_v, spans, nspans, lr, 0.9f, 0.999f, 1e-8f, grad_scale, s));
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

// ---- timing: HIP events on the launch stream around the kernels selected by `mask`
extern "C" int mfm_plan_set_timing(MfmPlan* P, int mask) {
  if (!P) return MFM_ERR_ARG;
  P->timing_mask = mask;
  return MFM_OK;
}
extern "C" int mfm_plan_num_kernels(void) { return K_COUNT; }
extern "C" const char* mfm_plan_kernel_name(int kid) {
  static const char* names[K_COUNT] = {"proj_gemm", "enc_seq_fwd", "latent_fwd", "dec_seq_fwd", "fc1_mse_gemm", "mse",
                                       "fc1_bwd_gemm", "dec_seq_bwd", "dec_dw_gemm", "latent_bwd", "enc_seq_bwd",
                                       "dw_gemm", "adam", "latent_dw_gemm", "bf16_weight_pack"};
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
  for (int e = 0; e < 4; ++e) {
    const double h = P->enc_h[e], d = P->enc_d[e];
    f += P->T * 2.0 * 4.0 * h * (d + h) + 2.0 * h * h;
  }
  for (int m = 0; m < 3; ++m) {
    const double h = P->dec_h[m], d = P->dec_d[m];
    f += P->T * (2.0 * 4.0 * h * (h + h) + 2.0 * h * d);
  }
  for (int i = 4; i < P->lat.nops; ++i) f += 2.0 * P->lat_ops[i].K * P->lat_ops[i].N;
  (void)c;
  return f;
}
extern "C" double mfm_plan_flops_per_step(const MfmPlan* P) { return P ? 3.0 * fwd_flops_per_sample(P) * P->B : 0.0; }
extern "C" double mfm_plan_bytes_per_step(const MfmPlan* P) {
  if (!P) return 0.0;
  double sh = 0.0;
  for (int e = 0; e < 4; ++e) sh += P->enc_h[e];
  for (int m = 0; m < 3; ++m) sh += P->dec_h[m];
  const double per_sample = 2.0 * P->T * P->D * 4.0 + 4.0 + 2.0 * P->T * 6.0 * sh * 4.0;
  return per_sample * P->B + 10.0 * (double)P->n_params * 4.0;
}
// Algorithmic FLOPs of ONE launch of kernel `kid` (recurrent/GEMM kernels only; 0 otherwise).
extern "C" double mfm_plan_kernel_flops(const MfmPlan* P, int kid) {
  if (!P) return 0.0;
  const double TB = (double)P->T * P->B;
  double f = 0.0;
  switch (kid) {
    case K_PROJ: for (int e = 0; e < 4; ++e) f += TB * 2.0 * 4.0 * P->enc_h[e] * P->enc_d[e]; break;
    case K_ENC_FWD: case K_ENC_BWD: for (int e = 0; e < 4; ++e) f += TB * 2.0 * 4.0 * P->enc_h[e] * P->enc_h[e]; break;
    case K_DEC_FWD: case K_DEC_BWD: for (int m = 0; m < 3; ++m) f += TB * 2.0 * 4.0 * P->dec_h[m] * P->dec_h[m]; break;
    case K_FC1_FWD: for (int m = 0; m < 3; ++m) f += TB * 2.0 * P->dec_h[m] * P->dec_d[m]; break;
    case K_FC1_BWD: for (int m = 0; m < 3; ++m) f += TB * 2.0 * P->dec_h[m] * P->dec_d[m]; break;   // dH only
    case K_DEC_DW: break;   // merged into K_ENC_DW
    case K_ENC_DW:
      for (int e = 0; e < 4; ++e) f += TB * 2.0 * 4.0 * P->enc_h[e] * (P->enc_d[e] + P->enc_h[e]);
      for (int m = 0; m < 3; ++m) f += TB * 2.0 * 4.0 * P->dec_h[m] * P->dec_h[m];
      for (int m = 0; m < 3; ++m) f += TB * 2.0 * P->dec_h[m] * P->dec_d[m];            // dWfc
      for (int i = 0; i < P->lat.nops; ++i) f += 2.0 * P->B * P->lat_ops[i].N * P->lat_ops[i].K;   // latent dW
      break;
    default: break;
  }
  return f;
}
