// Step-by-step LSTM sequence path for hidden sizes the weight-resident kernels do not take (h > 128).
//
// The reference's hyper-parameter search draws zl / fl / hl from {.., 156, 256} (mfm_mosi.py:1305-1320), so
// ef_encoder can be 416 wide and decoder_l 336: 4h^2 weights no longer fit one workgroup's registers (or
// LDS).  Those LSTMs run here exactly as the reference's loop does (mfm_model.py:47-58,72-88) -- one
// recurrent GEMM + one pointwise cell kernel per time step -- on the same buffers and with the same
// contract as lstm_seq.hip (gates: pre-activation -> activated -> dA in place, hs, cs, padded hidden Hp,
// zero pad units), so everything around them (input projection, fc1, weight-gradient GEMMs, the plan)
// is unchanged.  2T launches per direction instead of 1: a completeness path, not a fast one.
#include <stdlib.h>

#include "internal.h"
#include "lstm_seq_dev.h"

namespace mfm {

namespace {

struct CellItem {
  float* gates;            // [B,4,Hp] of step t
  const float* cprev;      // [B,Hp] or null (zeros)
  float* hs; float* cs;    // [B,Hp] of step t
  // backward only
  const float* cnow; const float* dh_rec; const float* ext; int64_t ld_ext; const float* dc_ext;
  float* dc;               // [B,Hp] carried cell-state gradient (in/out)
  int h, Hp, block_begin;
};
struct CellGroup { CellItem it[MFM_MAX_SEQ]; int count, B; };

__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(const CellGroup g) {
  int gi = 0;
#pragma unroll 1
  for (int i = 1; i < g.count; ++i)
    if ((int)blockIdx.x >= g.it[i].block_begin) gi = i;
  const CellItem& it = g.it[gi];
  const int Hp = it.Hp;
  const int64_t n = (int64_t)g.B * Hp;
  const int64_t idx = (int64_t)(blockIdx.x - it.block_begin) * 256 + threadIdx.x;
  if (idx >= n) return;
  const int64_t b = idx / Hp;
  const int u = (int)(idx - b * Hp);
  float* gp = it.gates + b * 4 * Hp + u;
  const float gi_ = act_sigmoid(gp[0]), gf = act_sigmoid(gp[Hp]), gg = act_tanh(gp[2 * Hp]), go = act_sigmoid(gp[3 * Hp]);
  const float cp = it.cprev ? it.cprev[idx] : 0.0f;
  const float c = fmaf(gf, cp, gi_ * gg);
  const float hv = go * act_tanh(c);
  gp[0] = gi_; gp[Hp] = gf; gp[2 * Hp] = gg; gp[3 * Hp] = go;
  it.cs[idx] = c;
  it.hs[idx] = (u < it.h) ? hv : 0.0f;
}

__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const CellGroup g) {
  int gi = 0;
#pragma unroll 1
  for (int i = 1; i < g.count; ++i)
    if ((int)blockIdx.x >= g.it[i].block_begin) gi = i;
  const CellItem& it = g.it[gi];
  const int Hp = it.Hp;
  const int64_t n = (int64_t)g.B * Hp;
  const int64_t idx = (int64_t)(blockIdx.x - it.block_begin) * 256 + threadIdx.x;
  if (idx >= n) return;
  const int64_t b = idx / Hp;
  const int u = (int)(idx - b * Hp);
  float* gp = it.gates + b * 4 * Hp + u;
  const float gi_ = gp[0], gf = gp[Hp], gg = gp[2 * Hp], go = gp[3 * Hp];
  const float ct = it.cnow[idx];
  const float cp = it.cprev ? it.cprev[idx] : 0.0f;
  float dh = it.dh_rec ? it.dh_rec[idx] : 0.0f;
  if (it.ext && (it.ld_ext == Hp || u < it.h)) dh += it.ext[b * it.ld_ext + u];
  const float dce = it.dc_ext ? it.dc_ext[idx] : 0.0f;
  const float tc = act_tanh(ct);
  const float dot = dh * tc;
  const float dct = dh * go * (1.0f - tc * tc) + it.dc[idx] + dce;
  const bool live = u < it.h;
  gp[0] = live ? dct * gg * gi_ * (1.0f - gi_) : 0.0f;
  gp[Hp] = live ? dct * cp * gf * (1.0f - gf) : 0.0f;
  gp[2 * Hp] = live ? dct * gi_ * (1.0f - gg * gg) : 0.0f;
  gp[3 * Hp] = live ? dot * go * (1.0f - go) : 0.0f;
  it.dc[idx] = dct * gf;
}

__global__ void add2_kernel(const float* a, const float* b, float* o, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) o[i] = a[i] + b[i];
}

int launch_cells(CellGroup& g, bool bwd, hipStream_t s) {
  int total = 0;
  for (int i = 0; i < g.count; ++i) {
    g.it[i].block_begin = total;
    total += (int)(((int64_t)g.B * g.it[i].Hp + 255) / 256);
  }
  if (bwd) MFM_LAUNCH_TIMED(lstm_cell_bwd_kernel, dim3(total), dim3(256), 0, s, g);
  else MFM_LAUNCH_TIMED(lstm_cell_fwd_kernel, dim3(total), dim3(256), 0, s, g);
  MFM_LAUNCH_CHECK(bwd ? "lstm_cell_bwd_kernel" : "lstm_cell_fwd_kernel");
  return MFM_OK;
}

// product over the 4 gate blocks:  C[B, 4, Hp] (+)= A[B, h] W[g]^T   (W [4h, h] row-major)
MfmGemmDesc rec_fwd_gemm(const float* a, int64_t lda, const float* w, float* gates, int B, int h, int Hp,
                         const float* b1, const float* b2, bool accumulate) {
  MfmGemmDesc d;
  memset(&d, 0, sizeof(d));
  d.a = a; d.a_sm = lda; d.a_sk = 1;
  d.b = w; d.b_sz = (int64_t)h * h; d.b_sn = h; d.b_sk = 1;
  d.c = gates; d.c_sz = Hp; d.ldc = 4 * (int64_t)Hp;
  d.bias = b1; d.bias2 = b2; d.bias_sz = h;
  d.m = B; d.n = Hp; d.n_valid = h; d.k = h; d.batch = 4; d.split_k = 1; d.accumulate = accumulate ? 1 : 0;
  d.alpha = 1.0f;
  return d;
}
// dH[B, ld] (+)= sum_g dA[B, g, :h] W[g]      (accumulating over the 4 gate blocks: C must be zeroed)
MfmGemmDesc rec_bwd_gemm(const float* gates, const float* w, float* out, int64_t ldo, int B, int h, int Hp) {
  MfmGemmDesc d;
  memset(&d, 0, sizeof(d));
  d.a = gates; d.a_sz = Hp; d.a_sm = 4 * (int64_t)Hp; d.a_sk = 1;
  d.b = w; d.b_sz = (int64_t)h * h; d.b_sk = h; d.b_sn = 1;
  d.c = out; d.c_sz = 0; d.ldc = ldo;
  d.m = B; d.n = h; d.n_valid = h; d.k = h; d.batch = 4; d.split_k = 1; d.accumulate = 1;
  d.alpha = 1.0f;
  return d;
}

}  // namespace

// descs: the LSTMs of one mfm_lstm_seq_fwd/bwd call whose h exceeds the resident kernels' limit
int seq_stepwise(const MfmSeqDesc* descs, int count, int T, int B, bool bwd, hipStream_t s) {
  MFM_REQUIRE(count >= 1 && count <= MFM_MAX_SEQ, "lstm_seq(stepwise): count %d", count);
  // scratch: W_ih + W_hh for decoders, carried dh / dc for the backward (stream-ordered allocation)
  size_t fl = 0;
  size_t off_ws[MFM_MAX_SEQ], off_dh[MFM_MAX_SEQ], off_dc[MFM_MAX_SEQ];
  for (int i = 0; i < count; ++i) {
    const MfmSeqDesc& d = descs[i];
    const size_t Hp = round_up(d.h, 16);
    off_ws[i] = fl; if (d.is_dec) fl += round_up64((int64_t)4 * d.h * d.h, 64);
    off_dh[i] = fl; if (bwd) fl += round_up64((int64_t)B * Hp, 64);
    off_dc[i] = fl; if (bwd) fl += round_up64((int64_t)B * Hp, 64);
  }
  float* scratch = nullptr;
  if (fl) MFM_HIP_CHECK(hipMallocAsync((void**)&scratch, fl * sizeof(float), s));
  int rc = MFM_OK;
  auto fail = [&](int code) { if (scratch) (void)hipFreeAsync(scratch, s); return code; };
  for (int i = 0; i < count; ++i) {
    const MfmSeqDesc& d = descs[i];
    if (d.is_dec) {
      const int64_t n = (int64_t)4 * d.h * d.h;
      MFM_LAUNCH_TIMED(add2_kernel, dim3((unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, s, d.w_ih, d.w_hh, scratch + off_ws[i], n);
    }
    if (bwd) {
      const size_t Hp = round_up(d.h, 16);
      if (hipMemsetAsync(scratch + off_dh[i], 0, 2 * round_up64((int64_t)B * Hp, 64) * sizeof(float), s) != hipSuccess) return fail(MFM_ERR_HIP);
    }
  }
  if (!bwd) {
    for (int t = 0; t < T; ++t) {
      MfmGemmDesc gd[MFM_MAX_SEQ];
      int ng = 0;
      CellGroup cg;
      memset(&cg, 0, sizeof(cg));
      cg.count = count; cg.B = B;
      for (int i = 0; i < count; ++i) {
        const MfmSeqDesc& d = descs[i];
        const int Hp = round_up(d.h, 16);
        float* gt = d.gates + (int64_t)t * B * 4 * Hp;
        if (d.is_dec) {
          if (t == 0) gd[ng++] = rec_fwd_gemm(d.h_init, d.ld_init, d.w_ih, gt, B, d.h, Hp, d.b_ih, d.b_hh, false);
          else gd[ng++] = rec_fwd_gemm(d.hs + (int64_t)(t - 1) * B * Hp, Hp, scratch + off_ws[i], gt, B, d.h, Hp, d.b_ih, d.b_hh, false);
        } else if (t > 0) {
          gd[ng++] = rec_fwd_gemm(d.hs + (int64_t)(t - 1) * B * Hp, Hp, d.w_hh, gt, B, d.h, Hp, nullptr, nullptr, true);
        }
        CellItem& it = cg.it[i];
        it.gates = gt;
        it.cprev = t > 0 ? d.cs + (int64_t)(t - 1) * B * Hp : nullptr;
        it.hs = d.hs + (int64_t)t * B * Hp; it.cs = d.cs + (int64_t)t * B * Hp;
        it.h = d.h; it.Hp = Hp;
      }
      if (ng) { rc = gemm_group_launch(gd, ng, s); if (rc != MFM_OK) return fail(rc); }
      rc = launch_cells(cg, false, s); if (rc != MFM_OK) return fail(rc);
    }
  } else {
    for (int t = T - 1; t >= 0; --t) {
      CellGroup cg;
      memset(&cg, 0, sizeof(cg));
      cg.count = count; cg.B = B;
      for (int i = 0; i < count; ++i) {
        const MfmSeqDesc& d = descs[i];
        const int Hp = round_up(d.h, 16);
        CellItem& it = cg.it[i];
        it.gates = d.gates + (int64_t)t * B * 4 * Hp;
        it.cnow = d.cs + (int64_t)t * B * Hp;
        it.cprev = t > 0 ? d.cs + (int64_t)(t - 1) * B * Hp : nullptr;
        it.dh_rec = scratch + off_dh[i];
        it.dc = scratch + off_dc[i];
        if (d.is_dec) { it.ext = d.dh_ext + (int64_t)t * B * d.ld_dh; it.ld_ext = d.ld_dh; }
        else if (t == T - 1) { it.ext = d.dh_ext; it.ld_ext = d.ld_dh; }
        it.dc_ext = d.dc_ext ? d.dc_ext + (int64_t)t * B * Hp : nullptr;
        it.h = d.h; it.Hp = Hp;
      }
      rc = launch_cells(cg, true, s); if (rc != MFM_OK) return fail(rc);
      // gradient wrt the recurrent input of this step
      MfmGemmDesc gd[MFM_MAX_SEQ];
      int ng = 0;
      for (int i = 0; i < count; ++i) {
        const MfmSeqDesc& d = descs[i];
        const int Hp = round_up(d.h, 16);
        const float* gt = d.gates + (int64_t)t * B * 4 * Hp;
        if (t > 0) {
          if (hipMemsetAsync(scratch + off_dh[i], 0, (size_t)B * Hp * sizeof(float), s) != hipSuccess) return fail(MFM_ERR_HIP);
          gd[ng++] = rec_bwd_gemm(gt, d.is_dec ? scratch + off_ws[i] : d.w_hh, scratch + off_dh[i], Hp, B, d.h, Hp);
        } else if (d.is_dec && d.d_h_init) {
          if (hipMemset2DAsync(d.d_h_init, d.ld_dinit * sizeof(float), 0, d.h * sizeof(float), B, s) != hipSuccess) return fail(MFM_ERR_HIP);
          gd[ng++] = rec_bwd_gemm(gt, d.w_ih, d.d_h_init, d.ld_dinit, B, d.h, Hp);
        }
      }
      if (ng) { rc = gemm_group_launch(gd, ng, s); if (rc != MFM_OK) return fail(rc); }
    }
  }
  if (scratch) MFM_HIP_CHECK(hipFreeAsync(scratch, s));
  return MFM_OK;
}

}  // namespace mfm
