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

#include "latent_row_dev.h"

namespace mfm {


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

// Register prefetch of a stage's weight span: the loads are issued one stage ahead (they overlap the
// previous stage's compute), the LDS writes happen when the panel is free.  Covers the first
// 8*nt chunks; copy_span handles a longer tail.
constexpr int PRE_U = 8;
// Buffer loads whose range is the span: a lane past the end reads 0 WITHOUT a memory request.  (The first version clamped
// the index instead: every stage then pulled 8 * nt * 16 bytes = 128 KB through the CU whatever the span's length --
// 0.9 MB instead of 228 KB per workgroup at ~10 B/clk, which was most of the staged kernels' time.)
__device__ __forceinline__ void span_load(const float* __restrict__ src, int n4, int tid, int nt, f32x4 (&pre)[PRE_U]) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n4 * 16, 0x00020000);
#pragma unroll
  for (int u = 0; u < PRE_U; ++u) pre[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (tid + u * nt) * 16, 0, 0));
}
__device__ __forceinline__ void span_store(const float* __restrict__ src, float* __restrict__ dst, int n4, int tid,
                                           int nt, const f32x4 (&pre)[PRE_U]) {
  f32x4* d4 = reinterpret_cast<f32x4*>(dst);
#pragma unroll
  for (int u = 0; u < PRE_U; ++u)
    if (tid + u * nt < n4) d4[tid + u * nt] = pre[u];
  if (n4 > PRE_U * nt) copy_span(src + 4 * PRE_U * nt, dst + 4 * PRE_U * nt, n4 - PRE_U * nt, tid, nt);
}


__device__ __forceinline__ void load_ops(const LatentDev& L, LatOp* ops) {
  for (int i = threadIdx.x; i < L.nops * (int)(sizeof(LatOp) / 4); i += blockDim.x)
    reinterpret_cast<int*>(ops)[i] = reinterpret_cast<const int*>(L.ops)[i];
}

// Per-stage exclusive prefix sums of the ops' N and K (filled in by the host, copied next to the op
// table).  find_op() then maps a work-item index to its op with <= 7 INDEPENDENT
// LDS reads and compares: a per-lane "while (local >= n) ++o" walk costs one dependent LDS round
// trip per step, ~2000 cycles per item in the 8-op stages.
__device__ __forceinline__ void build_prefix(const LatentDev& L, int* pfxN, int* pfxK) {
  for (int i = threadIdx.x; i < L.nops; i += blockDim.x) { pfxN[i] = L.ops[i].pfx_n; pfxK[i] = L.ops[i].pfx_k; }
}

// STAGED is a template parameter (not a runtime flag) so that the weight pointer has a static address
// space: a pointer that may be LDS or global compiles to flat_load + full waitcnt per access.
template <bool STAGED>
__global__ __launch_bounds__(LAT_THREADS) void latent_fwd_kernel(const LatentDev L, const float* __restrict__ params) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ LatOp ops[MFM_LAT_MAXOPS];
  __shared__ int pfxN[MFM_LAT_MAXOPS], pfxK[MFM_LAT_MAXOPS];
  __shared__ float red[2][16];
  load_ops(L, ops);
  build_prefix(L, pfxN, pfxK);
  float* rec = lds;
  const int RS = L.rec_size;
  const int R = L.rows_fwd;
  float* wp = lds + R * RS;
  constexpr bool staged = STAGED;
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
  f32x4 pre[PRE_U];
  if (staged) span_load(params + L.span_off[0], L.span_len[0] >> 2, tid, nt, pre);
  mark(L, 0);
  lds_barrier();     // record inputs + LDS op table
  mark(L, 1);

  for (int s = 0; s < L.nstages; ++s) {
    const int ob = L.stage_begin[s], oe = L.stage_begin[s + 1];
    const int64_t woff0 = staged ? L.span_off[s] : 0;   // weights addressed as base[w_off - woff0]
    if (staged) {
      span_store(params + L.span_off[s], wp, L.span_len[s] >> 2, tid, nt, pre);
      lds_barrier();     // LDS-only barrier: __syncthreads() would also drain the prefetch below (vmcnt(0))
      if (s + 1 < L.nstages) span_load(params + L.span_off[s + 1], L.span_len[s + 1] >> 2, tid, nt, pre);
    }
    mark(L, 2 + 2 * s);
    // one work item = (output column n of one op, k-quarter q, chunk of 4 batch rows): the four lanes of
    // a quad split the reduction dim in interleaved 16-byte chunks (ds_read_b128 for the weight row
    // and for each of the 4 activation rows: 5 LDS reads per 16 FMAs), all-reduce with two DPP adds,
    // then lane q finishes batch row q.  All 1024 threads have work (sum N x 4 ~ 1000 items).
    if (STAGED && L.mfma) {
      // MFMA form (round 3).  The quad form below reads 5 bytes of LDS per multiply-add (a 16-byte weight chunk and four
      // 16-byte activation chunks per 16 FMAs).  Here a wave owns a fragment of 16 output columns of one layer for ALL rows of the
      // workgroup (<= 16): D[row][n] += x[row][k] w[n][k] on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains), lane (bi, q)
      // supplying x[row bi][k] and w[n0 + bi][k] for k = 16 j + 4 q + e in MFMA step e of chunk j -- both operands of four
      // steps arrive as ONE 16-byte read each: 0.5 bytes of LDS per multiply-add.  Measured (B = 2048, profiles/r03_latent_mfma.txt):
      // 43.6 -> 39.4 us forward, 67.2 -> 60.0 us backward -- a stage's ~6000 cycles are mostly the serial bookkeeping
      // around the product (op look-up, bias, dropout stream, record writes: LDS round trips at one wave's pace), which
      // both forms share.
      const int lane = tid & 63, wave = tid >> 6, nwv = nt >> 6;
      const int bi = lane & 15, q = lane >> 4;
      int units = 0;
      for (int o = ob; o < oe; ++o) units += (ops[o].N + 15) >> 4;
      for (int u = wave; u < units; u += nwv) {
        int o = ob, f = u;
        while (f >= ((ops[o].N + 15) >> 4)) { f -= (ops[o].N + 15) >> 4; ++o; }
        const LatOp op = ops[o];
        const int n = f * 16 + bi;
        const float* ap = rec + min(bi, nrows - 1) * RS + op.in_off + 4 * q;
        const float* wq = wp + (op.w_off - woff0) + (int64_t)min(n, op.N - 1) * op.K + 4 * q;
        // four independent accumulation chains (a single one is 30 dependent MFMAs deep for K = 120: ~1500 cycles of
        // issue-to-result latency), two chunks per trip so that their four reads are in flight together
        f32x4 ac[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) ac[e] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kc = 0; kc < op.K; kc += 32) {
          const bool ok0 = kc + 4 * q < op.K, ok1 = kc + 16 + 4 * q < op.K;     // (K % 4 == 0: a chunk is whole or absent)
          const int k0 = ok0 ? kc : 0, k1 = ok1 ? kc + 16 : 0;
          f32x4 av0 = *reinterpret_cast<const f32x4*>(ap + k0), wv0 = *reinterpret_cast<const f32x4*>(wq + k0);
          f32x4 av1 = *reinterpret_cast<const f32x4*>(ap + k1), wv1 = *reinterpret_cast<const f32x4*>(wq + k1);
          if (!ok0) av0 = f32x4{0.f, 0.f, 0.f, 0.f};
          if (!ok1) av1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e) ac[e] = mma16x16x4(av0[e], wv0[e], ac[e]);
#pragma unroll
          for (int e = 0; e < 4; ++e) ac[e] = mma16x16x4(av1[e], wv1[e], ac[e]);
        }
        const f32x4 acc = (ac[0] + ac[1]) + (ac[2] + ac[3]);
        // accumulator register r of lane (bi, q): row 4q + r, column n
        if (n < op.N) {
          const float bv = wp[(op.b_off - woff0) + n];
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const int r = 4 * q + r4;
            if (r < nrows) {
              float v = acc[r4] + bv;
              if (op.relu) v = fmaxf(v, 0.0f);
              if (op.mask_off >= 0) {
                float mk = 1.0f;
                if (L.train && op.drop_p > 0.0f) {
                  const uint64_t idx = ((uint64_t)o << 40) + (uint64_t)(row0 + r) * (uint64_t)op.N + (uint64_t)n;
                  mk = (rng_uniform(L.seed + (L.tick ? *L.tick : 0ull), idx) < op.drop_p) ? 0.0f : 1.0f / (1.0f - op.drop_p);
                }
                v *= mk;
                rec[r * RS + op.mask_off + n] = mk;
              }
              rec[r * RS + op.out_off + n] = v;
            }
          }
        }
      }
      lds_barrier();
      mark(L, 3 + 2 * s);
      continue;
    }
    const int nch = (nrows + 3) >> 2;
    const int total = 4 * nch * (pfxN[oe - 1] + ops[oe - 1].N);
    for (int item4 = tid; item4 < ((total + 3) & ~3); item4 += nt) {
      const int q = item4 & 3;
      const int item = min(item4 >> 2, (total >> 2) - 1);
      const bool live = item4 < total;
      const int o = find_op(pfxN, ob, oe, item, nch);
      const int local = item - nch * pfxN[o];
      const LatOp op = ops[o];
      const int ch = (nch == 1) ? 0 : local / op.N, n = local - ch * op.N;
      const int r0 = ch * 4;
      const float* in[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) in[j] = rec + min(r0 + j, nrows - 1) * RS + op.in_off;
      const float* w;
      float bv;
      if constexpr (STAGED) {
        w = wp + (op.w_off - woff0) + (int64_t)n * op.K;
        bv = wp[(op.b_off - woff0) + n];
      } else {
        w = params + op.w_off + (int64_t)n * op.K;
        bv = params[op.b_off + n];
      }
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      if ((op.K & 3) == 0 && (STAGED || ((op.w_off & 3) == 0))) {
        for (int k = 4 * q; k < op.K; k += 16) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(in[j] + k);
            acc[j] = fmaf(xv[0], wv[0], acc[j]); acc[j] = fmaf(xv[1], wv[1], acc[j]);
            acc[j] = fmaf(xv[2], wv[2], acc[j]); acc[j] = fmaf(xv[3], wv[3], acc[j]);
          }
        }
      } else {
        for (int k = q; k < op.K; k += 4) {
          const float wv = w[k];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = fmaf(in[j][k], wv, acc[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = acc[j];
        v += dpp_quad<0xB1>(v);
        v += dpp_quad<0x4E>(v);
        acc[j] = v;
      }
      const float lo = (q & 1) ? acc[1] : acc[0], hi = (q & 1) ? acc[3] : acc[2];
      float v = ((q & 2) ? hi : lo) + bv;
      const int r = r0 + q;
      if (live && r < nrows) {
        if (op.relu) v = fmaxf(v, 0.0f);
        if (op.mask_off >= 0) {
          float mk = 1.0f;
          if (L.train && op.drop_p > 0.0f) {
            const uint64_t idx = ((uint64_t)o << 40) + (uint64_t)(row0 + r) * (uint64_t)op.N + (uint64_t)n;
            mk = (rng_uniform(L.seed + (L.tick ? *L.tick : 0ull), idx) < op.drop_p) ? 0.0f : 1.0f / (1.0f - op.drop_p);
          }
          v *= mk;
          rec[r * RS + op.mask_off + n] = mk;
        }
        rec[r * RS + op.out_off + n] = v;
      }
    }
    lds_barrier();
    mark(L, 3 + 2 * s);
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
  lds_barrier();
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

  // ---- outputs: plain stores, nothing in this kernel waits for them
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
  mark(L, 20);
}

template <bool STAGED>
__global__ __launch_bounds__(LAT_THREADS) void latent_bwd_kernel(const LatentDev L, const float* __restrict__ params,
                                                                 float* __restrict__ grads) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ LatOp ops[MFM_LAT_MAXOPS];
  __shared__ int pfxN[MFM_LAT_MAXOPS], pfxK[MFM_LAT_MAXOPS];
  load_ops(L, ops);
  build_prefix(L, pfxN, pfxK);
  const int RS = L.rec_size;
  const int R = L.rows_per_wg;
  float* rec = lds;
  float* grd = lds + R * RS;
  float* wp = lds + 2 * R * RS;
  constexpr bool staged = STAGED;
  const int row0 = blockIdx.x * R;
  const int nrows = min(R, L.B - row0);
  const int tid = threadIdx.x, nt = blockDim.x;

  {
    const int n4 = (nrows * RS) >> 2;
    const f32x4* s4 = reinterpret_cast<const f32x4*>(L.rec + (int64_t)row0 * RS);
    f32x4* r4 = reinterpret_cast<f32x4*>(rec);
    f32x4* g4 = reinterpret_cast<f32x4*>(grd);
    const f32x4* sd4 = L.grd_seed ? reinterpret_cast<const f32x4*>(L.grd_seed + (int64_t)row0 * RS) : nullptr;
    const float sw = sd4 ? (L.seed_w_ptr ? *L.seed_w_ptr : L.seed_w) : 0.0f;
    for (int idx = tid; idx < n4; idx += nt) { r4[idx] = s4[idx]; g4[idx] = sd4 ? sw * sd4[idx] : f32x4{0.f, 0.f, 0.f, 0.f}; }
  }
  lds_barrier();

  // ---- seeds
  if (L.d_yhat_ext) {
    for (int idx = tid; idx < nrows * L.od; idx += nt) {
      const int r = idx / L.od, o = idx - r * L.od;
      grd[r * RS + L.yhat_off + o] = L.d_yhat_ext[(int64_t)(row0 + r) * L.od + o];
    }
  } else if (L.y && (L.disc_w != 0.0f || L.disc_loss_out)) {
    float dl = 0.0f;                // this thread's share of the discriminative loss (disc_loss_out)
    if (L.loss_kind == 0) {
      const float* y = reinterpret_cast<const float*>(L.y);
      const float inv = 1.0f / ((float)L.B * (float)L.od);
      const float sc = L.disc_w * inv;
      for (int idx = tid; idx < nrows * L.od; idx += nt) {
        const int r = idx / L.od, o = idx - r * L.od;
        const float df = rec[r * RS + L.yhat_off + o] - y[(int64_t)(row0 + r) * L.od + o];
        grd[r * RS + L.yhat_off + o] = (df > 0.0f) ? sc : ((df < 0.0f) ? -sc : 0.0f);
        dl += fabsf(df) * inv;
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
        dl += ((logf(se) + mx) - z[lab]) / (float)L.B;
      }
    }
    if (L.disc_loss_out) {          // (uniform branch: every wave of the workgroup takes it)
      dl = wave_sum_dpp(dl);
      if ((tid & 63) == 0 && dl != 0.0f) atomicAdd(L.disc_loss_out, dl);
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
  f32x4 pre[PRE_U];
  if (staged) span_load(params + L.span_off[L.nstages - 1], L.span_len[L.nstages - 1] >> 2, tid, nt, pre);
  lds_barrier();
  mark(L, 24);

  for (int s = L.nstages - 1; s >= 0; --s) {
    const int ob = L.stage_begin[s], oe = L.stage_begin[s + 1];
    const int64_t woff0 = staged ? L.span_off[s] : 0;
    if (staged) {     // becomes visible at pass 1's barrier; the next stage's span is fetched meanwhile
      span_store(params + L.span_off[s], wp, L.span_len[s] >> 2, tid, nt, pre);
      if (s > 0) span_load(params + L.span_off[s - 1], L.span_len[s - 1] >> 2, tid, nt, pre);
    }
    // pass 1: gradient wrt the pre-activation, in place
    int total = nrows * (pfxN[oe - 1] + ops[oe - 1].N);
    for (int item = tid; item < total; item += nt) {
      const int o = find_op(pfxN, ob, oe, item, nrows);
      const int local = item - nrows * pfxN[o];
      const LatOp& op = ops[o];
      if (!op.relu && op.mask_off < 0) continue;
      const int r = local / op.N, n = local - r * op.N;
      float gv = grd[r * RS + op.out_off + n];
      if (op.relu && !(rec[r * RS + op.out_off + n] > 0.0f)) gv = 0.0f;
      if (op.mask_off >= 0) gv *= rec[r * RS + op.mask_off + n];
      grd[r * RS + op.out_off + n] = gv;
    }
    lds_barrier();       // also publishes the weight panel; LDS-only, the span prefetch stays in flight
    mark(L, 25 + 3 * s);
    // pass 2a: grad wrt the input segment (LDS atomics: several ops may share an input).
    // work item = (input column k of one op, n-quarter q, chunk of 4 rows): the quad splits the output
    // dim in interleaved chunks of 4, all-reduces with DPP, lane q adds batch row q.
    if (STAGED && L.mfma) {
      // MFMA form: a wave owns 16 input columns k of one layer for all rows: D[row][k] += g[row][n] w[n][k]; lane (bi, q)
      // supplies g[row bi][n] (one 16-byte read per four steps) and w[n][k0 + bi] (four 4-byte reads: a column of the
      // row-major weight) for n = 16 j + 4 q + e
      const int lane = tid & 63, wave = tid >> 6, nwv = nt >> 6;
      const int bi = lane & 15, q = lane >> 4;
      int units = 0;
      for (int o = ob; o < oe; ++o) units += (ops[o].K + 15) >> 4;
      for (int u = wave; u < units; u += nwv) {
        int o = ob, f = u;
        while (f >= ((ops[o].K + 15) >> 4)) { f -= (ops[o].K + 15) >> 4; ++o; }
        const LatOp op = ops[o];
        const int k = f * 16 + bi;
        const float* gp = grd + min(bi, nrows - 1) * RS + op.out_off + 4 * q;
        const float* wq = wp + (op.w_off - woff0) + min(k, op.K - 1);
        f32x4 ac[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) ac[e] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int nc = 0; nc < op.N; nc += 16) {
          const int n0 = nc + 4 * q;
          // (record segments are padded to multiples of 4 floats: a chunk that starts inside the segment can be read whole)
          const f32x4 gv = *reinterpret_cast<const f32x4*>(gp + (n0 < op.N ? nc : 0));
          float wv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) wv[e] = wq[min(n0 + e, op.N - 1) * op.K];
#pragma unroll
          for (int e = 0; e < 4; ++e) ac[e] = mma16x16x4(n0 + e < op.N ? gv[e] : 0.0f, wv[e], ac[e]);
        }
        const f32x4 acc = (ac[0] + ac[1]) + (ac[2] + ac[3]);
        if (k < op.K) {
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const int r = 4 * q + r4;
            if (r < nrows) atomicAdd(&grd[r * RS + op.in_off + k], acc[r4]);
          }
        }
      }
    } else {
    const int nch = (nrows + 3) >> 2;
    total = 4 * nch * (pfxK[oe - 1] + ops[oe - 1].K);
    for (int item4 = tid; item4 < ((total + 3) & ~3); item4 += nt) {
      const int q = item4 & 3;
      const int item = min(item4 >> 2, (total >> 2) - 1);
      const bool live = item4 < total;
      const int o = find_op(pfxK, ob, oe, item, nch);
      const int local = item - nch * pfxK[o];
      const LatOp op = ops[o];
      const int ch = (nch == 1) ? 0 : local / op.K, k = local - ch * op.K;
      const int r0 = ch * 4;
      const float* go[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) go[j] = grd + min(r0 + j, nrows - 1) * RS + op.out_off;
      const float* w;
      if constexpr (STAGED) w = wp + (op.w_off - woff0) + k; else w = params + op.w_off + k;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      if ((op.N & 3) == 0) {
        for (int n = 4 * q; n < op.N; n += 16) {
          float wv[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) wv[i] = w[(int64_t)(n + i) * op.K];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 gv = *reinterpret_cast<const f32x4*>(go[j] + n);
            acc[j] = fmaf(gv[0], wv[0], acc[j]); acc[j] = fmaf(gv[1], wv[1], acc[j]);
            acc[j] = fmaf(gv[2], wv[2], acc[j]); acc[j] = fmaf(gv[3], wv[3], acc[j]);
          }
        }
      } else {
        for (int n = q; n < op.N; n += 4) {
          const float wv = w[(int64_t)n * op.K];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = fmaf(go[j][n], wv, acc[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = acc[j];
        v += dpp_quad<0xB1>(v);
        v += dpp_quad<0x4E>(v);
        acc[j] = v;
      }
      const float lo = (q & 1) ? acc[1] : acc[0], hi = (q & 1) ? acc[3] : acc[2];
      const float v = (q & 2) ? hi : lo;
      if (live && r0 + q < nrows) atomicAdd(&grd[(r0 + q) * RS + op.in_off + k], v);
    }
    }
    mark(L, 26 + 3 * s);
    // pass 2b: bias gradients (column sums over this workgroup's rows).  The WEIGHT gradients
    // dW = G^T X are left to a grouped MFMA GEMM over the two records (plan.hip): writing 57k
    // floats per workgroup from here is store-issue bound (~7 B/clk/CU: 25 us at B=32), and
    // cross-XCD atomics on them were worse.
    total = L.skip_bias ? 0 : pfxN[oe - 1] + ops[oe - 1].N;      // (skip_bias: column sums of the weight-gradient launch, gemm_tn.hip)
    for (int item = tid; item < total; item += nt) {
      const int o = find_op(pfxN, ob, oe, item, 1);
      const LatOp& op = ops[o];
      const int n = item - pfxN[o];
      float a0 = 0.0f;
      for (int r = 0; r < nrows; ++r) a0 += grd[r * RS + op.out_off + n];
      atomicAdd(grads + op.b_off + n, a0);
    }
    lds_barrier();       // LDS-only: the bias atomics retire in the background (a full barrier waited ~4 us per stage for them)
    mark(L, 27 + 3 * s);
  }

  for (int m = 0; m < 4; ++m) {
    if (!L.dh_last[m]) continue;
    const int n = L.enc_n[m];
    for (int idx = tid; idx < nrows * n; idx += nt) {
      const int r = idx / n, k = idx - r * n;
      L.dh_last[m][(int64_t)(row0 + r) * L.dh_ld[m] + k] = grd[r * RS + L.in_off[m] + k];
    }
  }
  if (L.grd_out) {     // pre-activation gradients of every layer, for the dW GEMMs
    const int n4 = (nrows * RS) >> 2;
    f32x4* d4 = reinterpret_cast<f32x4*>(L.grd_out + (int64_t)row0 * RS);
    const f32x4* s4 = reinterpret_cast<const f32x4*>(grd);
    for (int idx = tid; idx < n4; idx += nt) d4[idx] = s4[idx];
  }
}

// The row kernels' bodies live in latent_row_dev.h (shared with the fold launches of lstm_seq_small.hip)
// the tails of the forward chains (mode 3 of the row body) as a launch of their own: what runs when the encoder launch left the
// tails behind and the decoder launch could not carry them after all
__global__ __launch_bounds__(LAT_THREADS) void latent_fwd_tail_kernel(const LatentDev L, const float* __restrict__ params) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int r = blockIdx.x;
  latent_fwd_row_body<false>(L, params, r % L.B, r / L.B, lds, true, 3);
}
int latent_fwd_tail_launch(const LatentDev& L, const float* params, hipStream_t stream) {
  const size_t lds1 = (size_t)latent_fwd_lds_floats(L.rec_size) * sizeof(float);
  MFM_HIP_CHECK(hipFuncSetAttribute((const void*)latent_fwd_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
  MFM_LAUNCH_TIMED(latent_fwd_tail_kernel, dim3(L.B * L.nch), dim3(LAT_THREADS), lds1, stream, L, params);
  MFM_LAUNCH_CHECK("latent_fwd_tail_kernel");
  return MFM_OK;
}

template <bool PRE>
__global__ __launch_bounds__(PRE ? LAT_PRE_THREADS : LAT_THREADS) void latent_fwd_row_kernel(const LatentDev L, const float* __restrict__ params) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int row = (int)blockIdx.x / L.nch, ch = (int)blockIdx.x - row * L.nch;
  latent_fwd_row_body<PRE>(L, params, row, ch, lds, false);
}

template <bool PRE>
__global__ __launch_bounds__(PRE ? LAT_PRE_THREADS : LAT_THREADS) void latent_bwd_row_kernel(const LatentDev L, const float* __restrict__ params,
                                                                     float* __restrict__ grads) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int row = (int)blockIdx.x / L.nch, ch = (int)blockIdx.x - row * L.nch;
  latent_bwd_row_body<PRE>(L, params, grads, row, ch, lds);
}

static int set_lds_limit(const void* fn, size_t bytes) {
  if (bytes > 64 * 1024) {
    MFM_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  }
  return MFM_OK;
}

int latent_fwd_launch(const LatentDev& L, const float* params, hipStream_t stream) {
  if (L.row_path) {
    const size_t lds1 = ((size_t)MFM_LAT_MAXSTAGES * MFM_LAT_ROW_THREADS * 4 + L.rec_size) * sizeof(float);
    if (L.pre) {
      if (int rc1 = set_lds_limit((const void*)latent_fwd_row_kernel<true>, lds1)) return rc1;
      MFM_LAUNCH_TIMED(latent_fwd_row_kernel<true>, dim3(L.B * L.nch), dim3(LAT_PRE_THREADS), lds1, stream, L, params);
    } else {
      if (int rc1 = set_lds_limit((const void*)latent_fwd_row_kernel<false>, lds1)) return rc1;
      MFM_LAUNCH_TIMED(latent_fwd_row_kernel<false>, dim3(L.B * L.nch), dim3(L.row_threads), lds1, stream, L, params);
    }
    MFM_LAUNCH_CHECK("latent_fwd_row_kernel");
    return MFM_OK;
  }
  const int R = L.rows_fwd;
  const size_t lds = ((size_t)R * L.rec_size + L.wpanel) * sizeof(float);
  MFM_REQUIRE(lds <= 156 * 1024, "latent_fwd: record + weight panel too large for LDS (%zu bytes)", lds);
  int rc = set_lds_limit(L.wpanel > 0 ? (const void*)latent_fwd_kernel<true> : (const void*)latent_fwd_kernel<false>, lds);
  if (rc != MFM_OK) return rc;
  if (L.wpanel > 0)
    MFM_LAUNCH_TIMED(latent_fwd_kernel<true>, dim3(cdiv(L.B, R)), dim3(LAT_THREADS), lds, stream, L, params);
  else
    MFM_LAUNCH_TIMED(latent_fwd_kernel<false>, dim3(cdiv(L.B, R)), dim3(LAT_THREADS), lds, stream, L, params);
  MFM_LAUNCH_CHECK("latent_fwd_kernel");
  return MFM_OK;
}
int latent_bwd_launch(const LatentDev& L, const float* params, float* grads, hipStream_t stream) {
  if (L.row_path) {
    const size_t lds1 = ((size_t)MFM_LAT_MAXSTAGES * MFM_LAT_ROW_THREADS * 4 + 2 * (size_t)L.rec_size) * sizeof(float);
    if (L.pre) {
      if (int rc1 = set_lds_limit((const void*)latent_bwd_row_kernel<true>, lds1)) return rc1;
      MFM_LAUNCH_TIMED(latent_bwd_row_kernel<true>, dim3(L.B * L.nch), dim3(LAT_PRE_THREADS), lds1, stream, L, params, grads);
    } else {
      if (int rc1 = set_lds_limit((const void*)latent_bwd_row_kernel<false>, lds1)) return rc1;
      MFM_LAUNCH_TIMED(latent_bwd_row_kernel<false>, dim3(L.B * L.nch), dim3(L.row_threads), lds1, stream, L, params, grads);
    }
    MFM_LAUNCH_CHECK("latent_bwd_row_kernel");
    return MFM_OK;
  }
  const int R = L.rows_per_wg;
  const size_t lds = (2 * (size_t)R * L.rec_size + L.wpanel) * sizeof(float);
  MFM_REQUIRE(lds <= 156 * 1024, "latent_bwd: records + weight panel too large for LDS (%zu bytes)", lds);
  int rc = set_lds_limit(L.wpanel > 0 ? (const void*)latent_bwd_kernel<true> : (const void*)latent_bwd_kernel<false>, lds);
  if (rc != MFM_OK) return rc;
  if (L.wpanel > 0)
    MFM_LAUNCH_TIMED(latent_bwd_kernel<true>, dim3(cdiv(L.B, R)), dim3(LAT_THREADS), lds, stream, L, params, grads);
  else
    MFM_LAUNCH_TIMED(latent_bwd_kernel<false>, dim3(cdiv(L.B, R)), dim3(LAT_THREADS), lds, stream, L, params, grads);
  MFM_LAUNCH_CHECK("latent_bwd_kernel");
  return MFM_OK;
}

}  // namespace mfm
