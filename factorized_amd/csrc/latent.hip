// Fused latent stack of MFM (reference mfm_model.py:630-647,657 + loss_KLD :36-38 + the
// discriminative loss of mfm_mosi.py:438 / mfm_you.py:484) in ONE kernel per direction.
//
// All 22 Linear layers here have fan-in/out <= ~120 and share one batch row; what costs time in
// the reference is 22 launches x (addmm + bias + relu + dropout) forward and ~70 autograd nodes
// backward.  Here a workgroup owns `rows_per_wg` batch rows, keeps every activation of those
// rows in an LDS "record" (one float segment per tensor), and walks a small op table stage by
// stage (ops inside a stage are independent).  The record is spilled once to HBM for the
// backward, which reloads it next to a gradient record of the same shape.
#include "internal.h"

namespace mfm {

__device__ __forceinline__ float wave_sum_l(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float dot_k(const float* __restrict__ in, const float* __restrict__ w, int K, bool vec) {
  float acc = 0.0f;
  if (vec) {
    for (int k = 0; k < K; k += 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(in + k);
      const f32x4 b = *reinterpret_cast<const f32x4*>(w + k);
      acc = fmaf(a[0], b[0], acc); acc = fmaf(a[1], b[1], acc);
      acc = fmaf(a[2], b[2], acc); acc = fmaf(a[3], b[3], acc);
    }
  } else {
    for (int k = 0; k < K; ++k) acc = fmaf(in[k], w[k], acc);
  }
  return acc;
}

__global__ __launch_bounds__(256) void latent_fwd_kernel(const LatentDev L, const float* __restrict__ params) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* rec = lds;
  const int RS = L.rec_size;
  const int R = L.rows_per_wg;
  const int row0 = blockIdx.x * R;
  const int nrows = min(R, L.B - row0);
  const int tid = threadIdx.x, nt = blockDim.x;

  for (int m = 0; m < 4; ++m) {
    const int n = L.enc_n[m];
    for (int idx = tid; idx < nrows * n; idx += nt) {
      const int r = idx / n, k = idx - r * n;
      rec[r * RS + L.in_off[m] + k] = L.enc_h[m][(int64_t)(row0 + r) * L.enc_ld[m] + k];
    }
  }
  __syncthreads();

  for (int s = 0; s < L.nstages; ++s) {
    const int ob = L.stage_begin[s], oe = L.stage_begin[s + 1];
    int total = 0;
    for (int o = ob; o < oe; ++o) total += nrows * L.op[o].N;
    for (int item = tid; item < total; item += nt) {
      int o = ob, local = item;
      while (local >= nrows * L.op[o].N) { local -= nrows * L.op[o].N; ++o; }
      const LatOp& op = L.op[o];
      const int r = local / op.N, n = local - r * op.N;
      const float* w = params + op.w_off + (int64_t)n * op.K;
      const bool vec = ((op.K & 3) == 0) && ((op.w_off & 3) == 0);
      float acc = params[op.b_off + n] + dot_k(rec + r * RS + op.in_off, w, op.K, vec);
      if (op.relu) acc = fmaxf(acc, 0.0f);
      if (op.mask_off >= 0) {
        float mk = 1.0f;
        if (L.train && op.drop_p > 0.0f) {
          const uint64_t idx = ((uint64_t)o << 40) + (uint64_t)(row0 + r) * (uint64_t)op.N + (uint64_t)n;
          mk = (rng_uniform(L.seed, idx) < op.drop_p) ? 0.0f : 1.0f / (1.0f - op.drop_p);
        }
        acc *= mk;
        rec[r * RS + op.mask_off + n] = mk;
      }
      rec[r * RS + op.out_off + n] = acc;
    }
    __syncthreads();
  }

  // ---- losses (partials per workgroup, one atomic each)
  float kld = 0.0f;
  if (L.has_logvar) {
    for (int m = 0; m < 4; ++m) {
      const int n = L.z_n[m];
      for (int idx = tid; idx < nrows * n; idx += nt) {
        const int r = idx / n, j = idx - r * n;
        const float mu = rec[r * RS + L.mu_off[m] + j], lv = rec[r * RS + L.lv_off[m] + j];
        kld += 1.0f + lv - mu * mu - expf(lv);
      }
    }
  }
  float disc = 0.0f;
  if (L.y) {
    if (L.loss_kind == 0) {
      const float* y = reinterpret_cast<const float*>(L.y);
      for (int idx = tid; idx < nrows * L.od; idx += nt) {
        const int r = idx / L.od, o = idx - r * L.od;
        disc += fabsf(rec[r * RS + L.yhat_off + o] - y[(int64_t)(row0 + r) * L.od + o]);
      }
    } else {
      const int64_t* y = reinterpret_cast<const int64_t*>(L.y);
      for (int r = tid; r < nrows; r += nt) {
        const float* z = rec + r * RS + L.yhat_off;
        float mx = z[0];
        for (int o = 1; o < L.od; ++o) mx = fmaxf(mx, z[o]);
        float se = 0.0f;
        for (int o = 0; o < L.od; ++o) se += expf(z[o] - mx);
        disc += (logf(se) + mx) - z[(int)y[row0 + r]];
      }
    }
  }
  __shared__ float red[2][4];
  kld = wave_sum_l(kld);
  disc = wave_sum_l(disc);
  if ((tid & 63) == 0) { red[0][tid >> 6] = kld; red[1][tid >> 6] = disc; }

  // ---- outputs
  const int fy = L.f_n[3];
  for (int m = 0; m < 3; ++m) {
    if (!L.dec_init[m]) continue;
    const int hd = fy + L.f_n[m];
    for (int idx = tid; idx < nrows * hd; idx += nt) {
      const int r = idx / hd, j = idx - r * hd;
      const float v = (j < fy) ? rec[r * RS + L.f_off[3] + j] : rec[r * RS + L.f_off[m] + (j - fy)];
      L.dec_init[m][(int64_t)(row0 + r) * L.dec_ld[m] + j] = v;
    }
  }
  if (L.yhat_out) {
    for (int idx = tid; idx < nrows * L.od; idx += nt) {
      const int r = idx / L.od, o = idx - r * L.od;
      L.yhat_out[(int64_t)(row0 + r) * L.od + o] = rec[r * RS + L.yhat_off + o];
    }
  }
  if (L.rec) {
    for (int idx = tid; idx < nrows * RS; idx += nt) L.rec[(int64_t)row0 * RS + idx] = rec[idx];
  }
  __syncthreads();
  if (tid == 0 && L.losses) {
    const int nw = (nt + 63) >> 6;
    float k = 0.0f, dsum = 0.0f;
    for (int i = 0; i < nw; ++i) { k += red[0][i]; dsum += red[1][i]; }
    if (L.has_logvar) atomicAdd(L.losses + 4, -0.5f * k);
    if (L.y) {
      const float inv = (L.loss_kind == 0) ? 1.0f / ((float)L.B * (float)L.od) : 1.0f / (float)L.B;
      atomicAdd(L.losses + 0, dsum * inv);
    }
  }
}

__global__ __launch_bounds__(256) void latent_bwd_kernel(const LatentDev L, const float* __restrict__ params,
                                                         float* __restrict__ grads) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int RS = L.rec_size;
  const int R = L.rows_per_wg;
  float* rec = lds;
  float* grd = lds + R * RS;
  const int row0 = blockIdx.x * R;
  const int nrows = min(R, L.B - row0);
  const int tid = threadIdx.x, nt = blockDim.x;

  for (int idx = tid; idx < nrows * RS; idx += nt) {
    rec[idx] = L.rec[(int64_t)row0 * RS + idx];
    grd[idx] = 0.0f;
  }
  __syncthreads();

  // ---- seeds
  if (L.d_yhat_ext) {
    for (int idx = tid; idx < nrows * L.od; idx += nt) {
      const int r = idx / L.od, o = idx - r * L.od;
      grd[r * RS + L.yhat_off + o] = L.d_yhat_ext[(int64_t)(row0 + r) * L.od + o];
    }
  } else if (L.y && L.disc_w != 0.0f) {
    if (L.loss_kind == 0) {
      const float* y = reinterpret_cast<const float*>(L.y);
      const float sc = L.disc_w / ((float)L.B * (float)L.od);
      for (int idx = tid; idx < nrows * L.od; idx += nt) {
        const int r = idx / L.od, o = idx - r * L.od;
        const float df = rec[r * RS + L.yhat_off + o] - y[(int64_t)(row0 + r) * L.od + o];
        grd[r * RS + L.yhat_off + o] = (df > 0.0f) ? sc : ((df < 0.0f) ? -sc : 0.0f);
      }
    } else {
      const int64_t* y = reinterpret_cast<const int64_t*>(L.y);
      const float sc = L.disc_w / (float)L.B;
      for (int r = tid; r < nrows; r += nt) {
        const float* z = rec + r * RS + L.yhat_off;
        float mx = z[0];
        for (int o = 1; o < L.od; ++o) mx = fmaxf(mx, z[o]);
        float se = 0.0f;
        for (int o = 0; o < L.od; ++o) se += expf(z[o] - mx);
        const int lab = (int)y[row0 + r];
        for (int o = 0; o < L.od; ++o)
          grd[r * RS + L.yhat_off + o] = sc * (expf(z[o] - mx) / se - (o == lab ? 1.0f : 0.0f));
      }
    }
  }
  if (L.gen_w != 0.0f) {
    const int fy = L.f_n[3];
    for (int idx = tid; idx < nrows * fy; idx += nt) {
      const int r = idx / fy, j = idx - r * fy;
      float s = 0.0f;
      for (int m = 0; m < 3; ++m)
        if (L.d_dec_init[m]) s += L.d_dec_init[m][(int64_t)(row0 + r) * L.dec_ld[m] + j];
      grd[r * RS + L.f_off[3] + j] = s;
    }
    for (int m = 0; m < 3; ++m) {
      if (!L.d_dec_init[m]) continue;
      const int n = L.f_n[m];
      for (int idx = tid; idx < nrows * n; idx += nt) {
        const int r = idx / n, j = idx - r * n;
        grd[r * RS + L.f_off[m] + j] = L.d_dec_init[m][(int64_t)(row0 + r) * L.dec_ld[m] + fy + j];
      }
    }
  }
  const float reg_w = L.reg_w_ptr ? *L.reg_w_ptr : L.reg_w;
  if (L.has_logvar && (L.reg_w_ptr || reg_w != 0.0f)) {
    for (int m = 0; m < 4; ++m) {
      const int n = L.z_n[m];
      for (int idx = tid; idx < nrows * n; idx += nt) {
        const int r = idx / n, j = idx - r * n;
        const float mu = rec[r * RS + L.mu_off[m] + j], lv = rec[r * RS + L.lv_off[m] + j];
        grd[r * RS + L.mu_off[m] + j] = reg_w * mu;
        grd[r * RS + L.lv_off[m] + j] = reg_w * (-0.5f) * (1.0f - expf(lv));
      }
    }
  }
  __syncthreads();

  for (int s = L.nstages - 1; s >= 0; --s) {
    const int ob = L.stage_begin[s], oe = L.stage_begin[s + 1];
    // pass 1: gradient wrt the pre-activation, in place
    int total = 0;
    for (int o = ob; o < oe; ++o) total += nrows * L.op[o].N;
    for (int item = tid; item < total; item += nt) {
      int o = ob, local = item;
      while (local >= nrows * L.op[o].N) { local -= nrows * L.op[o].N; ++o; }
      const LatOp& op = L.op[o];
      if (!op.relu && op.mask_off < 0) continue;
      const int r = local / op.N, n = local - r * op.N;
      float gv = grd[r * RS + op.out_off + n];
      if (op.relu && !(rec[r * RS + op.out_off + n] > 0.0f)) gv = 0.0f;
      if (op.mask_off >= 0) gv *= rec[r * RS + op.mask_off + n];
      grd[r * RS + op.out_off + n] = gv;
    }
    __syncthreads();
    // pass 2a: grad wrt the input segment (LDS atomics: several ops may share an input)
    total = 0;
    for (int o = ob; o < oe; ++o) total += nrows * L.op[o].K;
    for (int item = tid; item < total; item += nt) {
      int o = ob, local = item;
      while (local >= nrows * L.op[o].K) { local -= nrows * L.op[o].K; ++o; }
      const LatOp& op = L.op[o];
      const int r = local / op.K, k = local - r * op.K;
      const float* w = params + op.w_off + k;
      const float* go = grd + r * RS + op.out_off;
      float acc = 0.0f;
      for (int n = 0; n < op.N; ++n) acc = fmaf(go[n], w[(int64_t)n * op.K], acc);
      atomicAdd(&grd[r * RS + op.in_off + k], acc);
    }
    // pass 2b: parameter gradients, reduced over this workgroup's rows, one global atomic each
    total = 0;
    for (int o = ob; o < oe; ++o) total += L.op[o].N * (L.op[o].K + 1);
    for (int item = tid; item < total; item += nt) {
      int o = ob, local = item;
      while (local >= L.op[o].N * (L.op[o].K + 1)) { local -= L.op[o].N * (L.op[o].K + 1); ++o; }
      const LatOp& op = L.op[o];
      const int NK = op.N * op.K;
      float acc = 0.0f;
      if (local < NK) {
        const int n = local / op.K, k = local - n * op.K;
        for (int r = 0; r < nrows; ++r) acc = fmaf(grd[r * RS + op.out_off + n], rec[r * RS + op.in_off + k], acc);
        atomicAdd(grads + op.w_off + local, acc);
      } else {
        const int n = local - NK;
        for (int r = 0; r < nrows; ++r) acc += grd[r * RS + op.out_off + n];
        atomicAdd(grads + op.b_off + n, acc);
      }
    }
    __syncthreads();
  }

  for (int m = 0; m < 4; ++m) {
    if (!L.dh_last[m]) continue;
    const int n = L.enc_n[m];
    for (int idx = tid; idx < nrows * n; idx += nt) {
      const int r = idx / n, k = idx - r * n;
      L.dh_last[m][(int64_t)(row0 + r) * L.dh_ld[m] + k] = grd[r * RS + L.in_off[m] + k];
    }
  }
}

int latent_fwd_launch(const LatentDev& L, const float* params, hipStream_t stream) {
  const int R = L.rows_per_wg;
  const size_t lds = (size_t)R * L.rec_size * sizeof(float);
  MFM_REQUIRE(lds <= 64 * 1024, "latent_fwd: record too large for LDS (%zu bytes)", lds);
  hipLaunchKernelGGL(latent_fwd_kernel, dim3(cdiv(L.B, R)), dim3(256), lds, stream, L, params);
  MFM_LAUNCH_CHECK("latent_fwd_kernel");
  return MFM_OK;
}
int latent_bwd_launch(const LatentDev& L, const float* params, float* grads, hipStream_t stream) {
  const int R = L.rows_per_wg;
  const size_t lds = 2 * (size_t)R * L.rec_size * sizeof(float);
  MFM_REQUIRE(lds <= 64 * 1024, "latent_bwd: record too large for LDS (%zu bytes)", lds);
  hipLaunchKernelGGL(latent_bwd_kernel, dim3(cdiv(L.B, R)), dim3(256), lds, stream, L, params, grads);
  MFM_LAUNCH_CHECK("latent_bwd_kernel");
  return MFM_OK;
}

}  // namespace mfm
