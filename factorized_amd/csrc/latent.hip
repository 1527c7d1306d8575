// Fused latent stack of MFM (reference mfm_model.py:630-647,657 + loss_KLD :36-38 + the
// discriminative loss of mfm_mosi.py:438 / mfm_you.py:484) in ONE kernel per direction.
//
// All 22 Linear layers here have fan-in/out <= ~120 and share one batch row; what costs time in
// the reference is 22 launches x (addmm + bias + relu + dropout) forward and ~70 autograd nodes
// backward.  Here a workgroup owns `rows_per_wg` batch rows, keeps every activation of those
// rows in an LDS "record" (one float segment per tensor), and walks a small op table stage by
// stage (ops inside a stage are independent).  The record is spilled once to HBM for the
// backward, which reloads it next to a gradient record of the same shape.
//
// Weights.  The host lays the flat parameter buffer out so that the tensors of one stage are
// CONTIGUOUS (engine.py FlatLayout); a stage therefore starts with one linear, fully coalesced,
// 8-deep-unrolled copy of that span (<= ~93 KB at the canonical sizes) from L2 into an LDS panel:
// no per-tensor dependent round trips (the first versions paid one ~1 us round trip per tensor or
// per k-iteration: 61..127 us forward, 155..384 us backward at B=32, profiles/r01a).  Rows are NOT
// padded in LDS; instead the forward, whose lanes run over output columns n (stride K dwords,
// conflict-prone), walks k in a per-lane ROTATED order k' = (k + n) mod K, so a wave touches
// addresses n*(K+1)+k -- odd stride, bank-conflict free.  The backward's lanes run over k
// (stride 1) and need no rotation.
//
// The op table itself lives in device memory and is copied to LDS at kernel start: it is indexed
// with a per-lane op id, and a divergent index into the kernel-argument segment would make the
// compiler copy the whole table to scratch in every thread.
#include "internal.h"

namespace mfm {

constexpr int LAT_THREADS = 1024;

__device__ __forceinline__ float wave_sum_l(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// linear global -> LDS copy of `n4` 16-byte chunks, 8 loads in flight per thread
__device__ __forceinline__ void copy_span(const float* __restrict__ src, float* __restrict__ dst, int n4, int tid,
                                          int nt) {
  constexpr int U = 8;
  const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
  f32x4* d4 = reinterpret_cast<f32x4*>(dst);
  for (int base = tid; base < n4; base += nt * U) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = s4[min(base + u * nt, n4 - 1)];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (base + u * nt < n4) d4[base + u * nt] = v[u];
  }
}

__device__ __forceinline__ void load_ops(const LatentDev& L, LatOp* ops) {
  for (int i = threadIdx.x; i < L.nops * (int)(sizeof(LatOp) / 4); i += blockDim.x)
    reinterpret_cast<int*>(ops)[i] = reinterpret_cast<const int*>(L.ops)[i];
}

__global__ __launch_bounds__(LAT_THREADS) void latent_fwd_kernel(const LatentDev L, const float* __restrict__ params) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ LatOp ops[MFM_LAT_MAXOPS];
  __shared__ float red[2][16];
  load_ops(L, ops);
  float* rec = lds;
  const int RS = L.rec_size;
  const int R = L.rows_per_wg;
  float* wp = lds + R * RS;
  const bool staged = L.wpanel > 0;
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
  __syncthreads();   // record inputs + LDS op table

  for (int s = 0; s < L.nstages; ++s) {
    const int ob = L.stage_begin[s], oe = L.stage_begin[s + 1];
    const float* wbase = params;          // weights addressed as wbase[w_off - woff0]
    int64_t woff0 = 0;
    if (staged) {
      copy_span(params + L.span_off[s], wp, L.span_len[s] >> 2, tid, nt);
      __syncthreads();
      wbase = wp; woff0 = L.span_off[s];
    }
    // one item = one output column n of one op for a chunk of 4 batch rows: the weight row is
    // read once per 4 rows, the 4 accumulator chains are independent, lanes run over n.
    const int nch = (nrows + 3) >> 2;
    int total = 0;
    for (int o = ob; o < oe; ++o) total += nch * ops[o].N;
    for (int item = tid; item < total; item += nt) {
      int o = ob, local = item;
      while (local >= nch * ops[o].N) { local -= nch * ops[o].N; ++o; }
      const LatOp op = ops[o];
      const int ch = local / op.N, n = local - ch * op.N;
      const int r0 = ch * 4;
      const float* in[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) in[j] = rec + min(r0 + j, nrows - 1) * RS + op.in_off;
      const float* w = wbase + (op.w_off - woff0) + (int64_t)n * op.K;
      const float bv = wbase[(op.b_off - woff0) + n];
      float acc[4] = {bv, bv, bv, bv};
      int k = n % op.K;                    // rotated start: conflict-free LDS columns
#pragma unroll 4
      for (int kk = 0; kk < op.K; ++kk) {
        const float wv = w[k];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(in[j][k], wv, acc[j]);
        k = (k + 1 == op.K) ? 0 : k + 1;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = r0 + j;
        if (r >= nrows) continue;
        float v = acc[j];
        if (op.relu) v = fmaxf(v, 0.0f);
        if (op.mask_off >= 0) {
          float mk = 1.0f;
          if (L.train && op.drop_p > 0.0f) {
            const uint64_t idx = ((uint64_t)o << 40) + (uint64_t)(row0 + r) * (uint64_t)op.N + (uint64_t)n;
            mk = (rng_uniform(L.seed, idx) < op.drop_p) ? 0.0f : 1.0f / (1.0f - op.drop_p);
          }
          v *= mk;
          rec[r * RS + op.mask_off + n] = mk;
        }
        rec[r * RS + op.out_off + n] = v;
      }
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
    const int n4 = (nrows * RS) >> 2;
    f32x4* d4 = reinterpret_cast<f32x4*>(L.rec + (int64_t)row0 * RS);
    const f32x4* s4 = reinterpret_cast<const f32x4*>(rec);
    for (int idx = tid; idx < n4; idx += nt) d4[idx] = s4[idx];
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

__global__ __launch_bounds__(LAT_THREADS) void latent_bwd_kernel(const LatentDev L, const float* __restrict__ params,
                                                                 float* __restrict__ grads) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ LatOp ops[MFM_LAT_MAXOPS];
  load_ops(L, ops);
  const int RS = L.rec_size;
  const int R = L.rows_per_wg;
  float* rec = lds;
  float* grd = lds + R * RS;
  float* wp = lds + 2 * R * RS;
  const bool staged = L.wpanel > 0;
  const int row0 = blockIdx.x * R;
  const int nrows = min(R, L.B - row0);
  const int tid = threadIdx.x, nt = blockDim.x;

  {
    const int n4 = (nrows * RS) >> 2;
    const f32x4* s4 = reinterpret_cast<const f32x4*>(L.rec + (int64_t)row0 * RS);
    f32x4* r4 = reinterpret_cast<f32x4*>(rec);
    f32x4* g4 = reinterpret_cast<f32x4*>(grd);
    for (int idx = tid; idx < n4; idx += nt) { r4[idx] = s4[idx]; g4[idx] = f32x4{0.f, 0.f, 0.f, 0.f}; }
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
    const float* wbase = params;
    int64_t woff0 = 0;
    if (staged) {     // becomes visible at pass 1's barrier
      copy_span(params + L.span_off[s], wp, L.span_len[s] >> 2, tid, nt);
      wbase = wp; woff0 = L.span_off[s];
    }
    // pass 1: gradient wrt the pre-activation, in place
    int total = 0;
    for (int o = ob; o < oe; ++o) total += nrows * ops[o].N;
    for (int item = tid; item < total; item += nt) {
      int o = ob, local = item;
      while (local >= nrows * ops[o].N) { local -= nrows * ops[o].N; ++o; }
      const LatOp& op = ops[o];
      if (!op.relu && op.mask_off < 0) continue;
      const int r = local / op.N, n = local - r * op.N;
      float gv = grd[r * RS + op.out_off + n];
      if (op.relu && !(rec[r * RS + op.out_off + n] > 0.0f)) gv = 0.0f;
      if (op.mask_off >= 0) gv *= rec[r * RS + op.mask_off + n];
      grd[r * RS + op.out_off + n] = gv;
    }
    __syncthreads();
    // pass 2a: grad wrt the input segment (LDS atomics: several ops may share an input).
    // one item = one input column k of one op for a chunk of 4 rows; lanes run over k.
    const int nch = (nrows + 3) >> 2;
    total = 0;
    for (int o = ob; o < oe; ++o) total += nch * ops[o].K;
    for (int item = tid; item < total; item += nt) {
      int o = ob, local = item;
      while (local >= nch * ops[o].K) { local -= nch * ops[o].K; ++o; }
      const LatOp op = ops[o];
      const int ch = local / op.K, k = local - ch * op.K;
      const int r0 = ch * 4;
      const float* go[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) go[j] = grd + min(r0 + j, nrows - 1) * RS + op.out_off;
      const float* w = wbase + (op.w_off - woff0) + k;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int n = 0; n < op.N; ++n) {
        const float wv = w[(int64_t)n * op.K];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(go[j][n], wv, acc[j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (r0 + j < nrows) atomicAdd(&grd[(r0 + j) * RS + op.in_off + k], acc[j]);
    }
    // pass 2b: parameter gradients, reduced over this workgroup's rows, one global atomic each
    total = 0;
    for (int o = ob; o < oe; ++o) total += ops[o].N * (ops[o].K + 1);
    for (int item = tid; item < total; item += nt) {
      int o = ob, local = item;
      while (local >= ops[o].N * (ops[o].K + 1)) { local -= ops[o].N * (ops[o].K + 1); ++o; }
      const LatOp op = ops[o];
      const int NK = op.N * op.K;
      float a0 = 0.0f, a1 = 0.0f;
      if (local < NK) {
        const int n = local / op.K, k = local - n * op.K;
        const float* g = grd + op.out_off + n;
        const float* x = rec + op.in_off + k;
        int r = 0;
        for (; r + 1 < nrows; r += 2) {
          a0 = fmaf(g[r * RS], x[r * RS], a0);
          a1 = fmaf(g[(r + 1) * RS], x[(r + 1) * RS], a1);
        }
        if (r < nrows) a0 = fmaf(g[r * RS], x[r * RS], a0);
        atomicAdd(grads + op.w_off + local, a0 + a1);
      } else {
        const int n = local - NK;
        for (int r = 0; r < nrows; ++r) a0 += grd[r * RS + op.out_off + n];
        atomicAdd(grads + op.b_off + n, a0);
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

static int set_lds_limit(const void* fn, size_t bytes) {
  if (bytes > 64 * 1024) {
    MFM_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  }
  return MFM_OK;
}

int latent_fwd_launch(const LatentDev& L, const float* params, hipStream_t stream) {
  const int R = L.rows_per_wg;
  const size_t lds = ((size_t)R * L.rec_size + L.wpanel) * sizeof(float);
  MFM_REQUIRE(lds <= 156 * 1024, "latent_fwd: record + weight panel too large for LDS (%zu bytes)", lds);
  int rc = set_lds_limit((const void*)latent_fwd_kernel, lds);
  if (rc != MFM_OK) return rc;
  hipLaunchKernelGGL(latent_fwd_kernel, dim3(cdiv(L.B, R)), dim3(LAT_THREADS), lds, stream, L, params);
  MFM_LAUNCH_CHECK("latent_fwd_kernel");
  return MFM_OK;
}
int latent_bwd_launch(const LatentDev& L, const float* params, float* grads, hipStream_t stream) {
  const int R = L.rows_per_wg;
  const size_t lds = (2 * (size_t)R * L.rec_size + L.wpanel) * sizeof(float);
  MFM_REQUIRE(lds <= 156 * 1024, "latent_bwd: records + weight panel too large for LDS (%zu bytes)", lds);
  int rc = set_lds_limit((const void*)latent_bwd_kernel, lds);
  if (rc != MFM_OK) return rc;
  hipLaunchKernelGGL(latent_bwd_kernel, dim3(cdiv(L.B, R)), dim3(LAT_THREADS), lds, stream, L, params, grads);
  MFM_LAUNCH_CHECK("latent_bwd_kernel");
  return MFM_OK;
}

}  // namespace mfm
