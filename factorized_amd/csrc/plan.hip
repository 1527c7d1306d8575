// The fused plan, C ABI (include/mfm_hip.h): create / destroy / workspace, per-plan switches and device-side state, the step
// entry points (forward, backward in its three forms, grad_step, train_step, train_step_staged), layout queries, timing.
// The chains themselves: plan_forward.hip, plan_backward.hip, plan_mfn.hip; tables: plan_build.hip.
#include "plan_internal.h"
#include "lstamp.h"

using namespace mfm;

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
                                       "fc1_bwd_gemm", "dec_seq_bwd", "lstm_dw_stream", "latent_bwd", "enc_seq_bwd",
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
  // SURVEY.md section 8d per sample: the batch read twice, the label, the saved LSTM state (i, f, g, o, c, h per unit and time
  // step) written once and read once; per step: parameter traffic (forward read, backward read, gradient write, Adam 4 R + 3 W).
  // bf16-RESIDENT plans (round 6: the fp32 formula over-stated them, 896 MB where the layout moves ~575 MB at B = 2048) keep the
  // gates and the hidden states as bf16 (5 of the 6 saved values per unit: the cell state stays fp32), read the batch once as fp32
  // and once as its bf16 image, and add the bf16 d x_hat / dH streams of the decoders (written and read once each)
  double per_sample;
  if (P->st16) {
    double dd = 0.0, dh = 0.0;
    for (int m = 0; m < 3; ++m) { dd += P->dec_d[m]; dh += P->dec_h[m]; }
    per_sample = P->T * P->D * (4.0 + 2.0 + 2.0) + 4.0 + 2.0 * P->T * sh * (5.0 * 2.0 + 4.0) + 2.0 * P->T * (dd + dh) * 2.0;
  } else {
    per_sample = 2.0 * P->T * P->D * 4.0 + 4.0 + 2.0 * P->T * 6.0 * sh * 4.0;
  }
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

#if MFM_LAUNCH_STAMP
// stamp build only (lstamp.h): the device buffer of the launch clock and its read-out (not part of the product ABI)
namespace mfm {
unsigned long long* lstamp_buffer() {
  static unsigned long long* buf = nullptr;
  if (!buf) {
    const size_t bytes = (size_t)LST_KERNELS * LST_BLOCKS * LST_POINTS * sizeof(unsigned long long);
    if (hipMalloc(&buf, bytes) != hipSuccess) return nullptr;
    (void)hipMemset(buf, 0, bytes);
  }
  return buf;
}
}  // namespace mfm
extern "C" int mfm_debug_lstamp_dims(int* kernels, int* blocks, int* points) {
  *kernels = LST_KERNELS; *blocks = LST_BLOCKS; *points = LST_POINTS;
  return MFM_OK;
}
// copies the stamps to `dst` (host) and clears the device buffer; synchronises the device
extern "C" int mfm_debug_lstamp_read(unsigned long long* dst) {
  unsigned long long* b = lstamp_buffer();
  if (!b || !dst) return MFM_ERR_ARG;
  const size_t bytes = (size_t)LST_KERNELS * LST_BLOCKS * LST_POINTS * sizeof(unsigned long long);
  MFM_HIP_CHECK(hipDeviceSynchronize());
  MFM_HIP_CHECK(hipMemcpy(dst, b, bytes, hipMemcpyDeviceToHost));
  MFM_HIP_CHECK(hipMemset(b, 0, bytes));
  return MFM_OK;
}
#endif
