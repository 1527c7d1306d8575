// Device bodies of the latent-stack ROW kernels (one batch row, or one modality chain of a row, per workgroup), shared by
// latent.hip (their own launches) and lstm_seq_small.hip (the fold launches: an encoder recurrence workgroup runs its
// row's chain right behind its last time step, and ahead of its BPTT).  Test infrastructure does not include this file.
#pragma once
#include "internal.h"
#include "lstamp.h"

namespace mfm {

constexpr int LAT_THREADS = 1024;
constexpr int LAT_PRE_THREADS = 512;         // chain workgroups with every stage's weights requested up front (256 VGPRs)
constexpr int LAT_PRE_STAGES = 6;
constexpr int LAT_PRE_SLOTS = 4;             // weights requested 4 stages ahead (6 resident slots spill: 222 + 60 registers)

template <int CTRL>
__device__ __forceinline__ float dpp_quad(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}


// debug: phase timestamps of workgroup 0 (shader clock), enabled with MFM_LATENT_DBG=1
__device__ __forceinline__ void mark(const LatentDev& L, int slot) {
  if (L.dbg && blockIdx.x == 0 && threadIdx.x == 0) L.dbg[slot] = __builtin_readcyclecounter();
}

__device__ __forceinline__ int find_op(const int* pfx, int ob, int oe, int x, int mult) {
  int o = ob;
#pragma unroll
  for (int i = 1; i < 8; ++i) {
    const int idx = min(ob + i, oe - 1);
    const int p = pfx[idx] * mult;
    o += (int)((ob + i < oe) & (x >= p));
  }
  return o;
}

// =====================================================================================================
// Latency path: ONE batch row per workgroup, weights straight from L2 into registers.
//
// At the reference's minibatch (B=32) the stack is a chain of dependent stages of tiny matvecs; what costs
// time is the chain, not the arithmetic.  The staged kernels above put 4 rows in a workgroup (8 workgroups
// at B=32), copy each stage's weights into LDS and walk k in a dependent loop: ~4.5 us per stage
// (profiles/r01 latent phase timeline).  Here every row gets its own workgroup (B workgroups), each weight
// element is used exactly once per workgroup, so it goes global -> register with all loads of a stage in
// flight at once, and the only LDS traffic is the activation / gradient record:
//   forward : lane (n, q) of a quad holds W[n][4q+16j .. +3], j < 8, dots it with the input segment,
//             quad all-reduce with two DPP adds.
//   backward: lane (kc, l) of a 16-lane group holds W[l+16j][4kc .. +3], j < 8, accumulates g[n]*W[n][k],
//             16-lane all-reduce with four DPP adds per value, lanes 0..3 add dX[4kc+l] into the record.
// Which (layer, column) a thread owns in a stage is static; the host tabulates it (plan.hip) as one int4 per
// thread and stage, the prologue copies the table into LDS, so a stage starts with one LDS read instead of
// a search through the op table:
//   forward  x = element offset of weight row n          y = element offset of bias[n]
//            z = in_off | K << 16                         w = (out_off + n) | op << 16 | relu << 24 | mask << 25 | live << 26
//   backward x = element offset of W[0][kc]               y = K | N << 8
//            z = out_off | (in_off + kc) << 16            w = live | producer relu << 1 | (producer mask index + 1) << 2
// Weights of stage s+1 are requested while stage s computes (two register slots whose roles swap by
// unrolling the stage loop twice: copying a slot would need the data, i.e. wait for the prefetch).
// Requirements (checked on the host, otherwise the staged kernels run): K % 4 == 0, K, N <= 128, weights
// 16-byte aligned, one work item per thread and stage (4*sum N, 4*sum K <= 1024), B <= 256.

template <int CTRL>
__device__ __forceinline__ float dpp_row(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}
// sum over the 64 lanes of a wave, result valid in every lane's return value only for lane groups' leaders:
// 16-lane all-reduce with DPP, then the four row leaders are combined through scalar registers
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_row<0xB1>(v); v += dpp_row<0x4E>(v);
  v += dpp_row<0x141>(v); v += dpp_row<0x140>(v);
  const int iv = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
}

typedef int i32x4 __attribute__((ext_vector_type(4)));

// copy this thread's items of all stages (one coalesced 16-byte load each) into the LDS table
__device__ __forceinline__ void load_items(const int* __restrict__ items, int nstages, i32x4* tab, int tid) {
  const i32x4* src = reinterpret_cast<const i32x4*>(items);
  i32x4 v[MFM_LAT_MAXSTAGES];
#pragma unroll
  for (int s = 0; s < MFM_LAT_MAXSTAGES; ++s) v[s] = src[min(s, nstages - 1) * MFM_LAT_ROW_THREADS + tid];
#pragma unroll
  for (int s = 0; s < MFM_LAT_MAXSTAGES; ++s) tab[s * MFM_LAT_ROW_THREADS + tid] = v[s];
}

// PRE (chain workgroups, <= 6 stages, <= 512 work items per stage): 512 threads, and the weights of ALL stages are requested
// before the first one runs -- a chain is 4-6 dependent stages of almost no arithmetic, and with one-stage-ahead
// prefetch every stage still costs the ~2 us its cold weights take to arrive (the optimizer rewrote them a step ago).
// `row`, `ch`: the batch row and (L.nch == 4) the modality chain of this workgroup.  `own_input`: the workgroup is the
// encoder recurrence of (row, ch) that has just written its last hidden state (fold launch, lstm_seq_small.hip): only
// that input may be read -- the other encoders' workgroups may still be running.
// `mode` (round 6; the launch clock put 3.3 us between the last time step and the first stage: the 96 KB item table, the op
// table and a re-read of the row's own h_T from global memory behind a store wait): 0 = the whole body; 1 = PRELOAD only --
// everything of the prologue that depends on nothing of this launch (op table, item table, target) goes to LDS and returns, no
// barrier: the fold launch calls it before the time loop, whose buffers lie behind this body's LDS region
// (latent_fwd_lds_floats); 2 = the rest, with the chain's input taken from `h_lds` (the recurrence's last hidden state, LDS).
// `stage_hi` (mode 2, round 6): stop behind stage stage_hi - 1 -- the decoders' inputs and the chain's part of the record go to
// memory, the classifier / logvar stages, the losses and y_hat are left to mode 3: a TAIL block of the decoder launch (any idle
// CU there; lstm_seq_small_dectail_kernel) that reloads the record and finishes the chain while the decoders run.  The launch
// clock put those stages + losses at 2.4 us of the encoder launch's critical path; nothing in the decoder launch waits for them.
__host__ __device__ static inline int latent_fwd_lds_floats(int rec_size) { return MFM_LAT_MAXSTAGES * MFM_LAT_ROW_THREADS * 4 + rec_size; }
template <bool PRE>
__device__ __forceinline__ void latent_fwd_row_body(const LatentDev& L, const float* __restrict__ params, const int row, const int ch,
                                                    float* lds, const bool own_input, const int mode = 0,
                                                    const float* h_lds = nullptr, const int stage_hi = MFM_LAT_MAXSTAGES) {
#if MFM_LAUNCH_STAMP
  const int LST_KF = mode == 3 ? 1 : 0;        // a tail block is a block of the decoder forward launch
#endif
  __shared__ LatOp ops[MFM_LAT_MAXOPS];
  __shared__ float red[2][16];
  __shared__ float ysave[2][128];
  i32x4* tab = reinterpret_cast<i32x4*>(lds);                        // [MAXSTAGES][1024] items
  float* rec = lds + MFM_LAT_MAXSTAGES * MFM_LAT_ROW_THREADS * 4;
  // nch == 4: this workgroup runs ONE modality chain (l, a, v or y) of its row -- the chains are independent inside the
  // stack, and four CUs stream a row's weights instead of one (LatentDev::nch)
  const int nch = L.nch;
  const bool all = nch == 1;
  const int tid = threadIdx.x, nt = blockDim.x;
  int nif[MFM_LAT_MAXSTAGES];
#pragma unroll
  for (int i = 0; i < MFM_LAT_MAXSTAGES; ++i) nif[i] = L.nitems_fwd_c[ch][i];
  // prologue: op table, item table and the four encoder states are requested together (one round trip)
  float yv = 0.0f;
  int ylab = 0;
  {
    const int nw = L.nops * (int)(sizeof(LatOp) / 4);
    const int e0 = L.enc_n[0], e1 = e0 + L.enc_n[1], e2 = e1 + L.enc_n[2], e3 = e2 + L.enc_n[3];
    const int tt = min(tid, e3 - 1);
    const int m = (tt >= e0) + (tt >= e1) + (tt >= e2);
    const int kk = tt - (m == 0 ? 0 : (m == 1 ? e0 : (m == 2 ? e1 : e2)));
    const int io = m == 0 ? L.in_off[0] : (m == 1 ? L.in_off[1] : (m == 2 ? L.in_off[2] : L.in_off[3]));
    const bool in_ok = !own_input || m == ch;
    if (mode == 3) {        // tail block: tables, target, and this chain's part of the record as the fold launch left it
      const int opv = reinterpret_cast<const int*>(L.ops)[min(tid, nw - 1)];
      if (L.y) {
        if (L.loss_kind == 0) yv = reinterpret_cast<const float*>(L.y)[(int64_t)row * L.od + min(tid, L.od - 1)];
        else ylab = (int)reinterpret_cast<const int64_t*>(L.y)[row];
      }
      const int lo4 = all ? 0 : (L.ch_lo[ch] >> 2), hi4 = all ? (L.rec_size >> 2) : (L.ch_hi[ch] >> 2);
      const f32x4* s4 = reinterpret_cast<const f32x4*>(L.rec + (int64_t)row * L.rec_size);
      f32x4* r4 = reinterpret_cast<f32x4*>(rec);
      const f32x4 rv = s4[min(lo4 + tid, hi4 - 1)];
      load_items(L.items_fwd + (size_t)ch * (MFM_LAT_MAXSTAGES * MFM_LAT_ROW_THREADS * 4), L.nstages, tab, tid);
      if (tid < nw) reinterpret_cast<int*>(ops)[tid] = opv;
      if (lo4 + tid < hi4) r4[lo4 + tid] = rv;
      for (int idx = lo4 + tid + nt; idx < hi4; idx += nt) r4[idx] = s4[idx];
    } else if (mode != 2) {
      const int opv = reinterpret_cast<const int*>(L.ops)[min(tid, nw - 1)];
      const float* src = m == 0 ? L.enc_h[0] : (m == 1 ? L.enc_h[1] : (m == 2 ? L.enc_h[2] : L.enc_h[3]));
      const int64_t ld = m == 0 ? L.enc_ld[0] : (m == 1 ? L.enc_ld[1] : (m == 2 ? L.enc_ld[2] : L.enc_ld[3]));
      float hv = 0.0f;
      if (mode == 0) hv = in_ok ? src[(int64_t)row * ld + kk] : 0.0f;
      if (L.y) {      // the target is needed only by the loss at the very end: fetch it now, not there
        if (L.loss_kind == 0) yv = reinterpret_cast<const float*>(L.y)[(int64_t)row * L.od + min(tid, L.od - 1)];
        else ylab = (int)reinterpret_cast<const int64_t*>(L.y)[row];
      }
      load_items(L.items_fwd + (size_t)ch * (MFM_LAT_MAXSTAGES * MFM_LAT_ROW_THREADS * 4), L.nstages, tab, tid);
      if (tid < nw) reinterpret_cast<int*>(ops)[tid] = opv;
      if (mode == 0 && tid < e3) rec[io + kk] = hv;
      if (mode == 1) {
        if (tid < 128) { ysave[0][tid] = yv; ysave[1][tid] = __builtin_bit_cast(float, ylab); }
        return;
      }
    } else {
      if (tid < e3) rec[io + kk] = in_ok ? h_lds[kk] : 0.0f;
      yv = ysave[0][min(tid, 127)]; ylab = __builtin_bit_cast(int, ysave[1][0]);
    }
  }
  // Descriptor fields the epilogue needs are fetched NOW (scalar loads, first touch of those kernarg lines)
  // and pinned in SGPRs: read where they are used they cost the tail of the kernel a chain of cold misses.
#define PIN_S(x) asm volatile("" : "+s"(x))
  int e_mu[4], e_lv[4], e_zn[4], e_fo[4], e_fn[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    e_mu[m] = L.mu_off[m]; e_lv[m] = L.lv_off[m]; e_zn[m] = L.z_n[m]; e_fo[m] = L.f_off[m]; e_fn[m] = L.f_n[m];
    PIN_S(e_mu[m]); PIN_S(e_lv[m]); PIN_S(e_zn[m]); PIN_S(e_fo[m]); PIN_S(e_fn[m]);
  }
  float* e_dec[3]; int64_t e_ld[3];
#pragma unroll
  for (int m = 0; m < 3; ++m) { e_dec[m] = L.dec_init[m]; e_ld[m] = L.dec_ld[m]; PIN_S(e_dec[m]); PIN_S(e_ld[m]); }
  int e_yoff = L.yhat_off, e_od = L.od, e_rs = L.rec_size, e_B = L.B, e_kind = L.loss_kind, e_haslv = L.has_logvar;
  float* e_yout = L.yhat_out; float* e_rec = L.rec; float* e_losses = L.losses;
  PIN_S(e_yoff); PIN_S(e_od); PIN_S(e_rs); PIN_S(e_B); PIN_S(e_kind); PIN_S(e_haslv); PIN_S(e_yout); PIN_S(e_rec); PIN_S(e_losses);
#undef PIN_S
  const int q = tid & 3;
  const int wave0 = tid & ~63;
  struct Slot { f32x4 w[8]; float bias; i32x4 e; };
  // (round 6) The weight requests are buffer loads: a k-block beyond the layer's K is asked for at an offset outside the
  // parameter buffer -- it still counts as one of the stage's eight requests (the counted waits stay exact) but returns zeros
  // without touching the cache.  Clamped re-reads of the last block used to occupy the CU's address path like real ones: a stage
  // whose layers have K = 32 issued four times the requests it needed, and the request issue is most of a stage.
  const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)params, 0, (int)min((int64_t)0x7fffffff, L.n_params * 4), 0x00020000);
  auto fetch = [&](int s, Slot& t) {          // weights + bias of this thread's item in stage s
    t.e = tab[s * MFM_LAT_ROW_THREADS + tid];
    const int K = (t.e[2] >> 16) & 0xFF;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k0 = 4 * q + 16 * j;
      t.w[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, k0 < K ? (int)(((unsigned)t.e[0] + k0) * 4u) : 0x7FFFFFF0, 0, 0));
    }
    t.bias = params[(unsigned)t.e[1]];
  };
  mark(L, 0);
  lds_barrier();
  mark(L, 1);
  LSTAMP(LST_KF, 16);
  auto stage = [&](int s, Slot& cur, Slot& nxt) {
    // A wave with no item in this stage nor in the next skips the body.  Inside the body every load is
    // unconditional (the last stage requests its own weights again): a load under a branch would make the
    // compiler's in-order vmcnt accounting conservative and the wait for `cur` would also cover `nxt`.
    const int sn = min(s + 1, L.nstages - 1);
#if MFM_LAUNCH_STAMP
    if (s == 2) LSTAMP(LST_KF, 26);
#endif
    if (wave0 < (PRE ? nif[s] : max(nif[s], nif[sn]))) {
      const int in_off = cur.e[2] & 0xFFFF, K = (cur.e[2] >> 16) & 0xFF;
      f32x4 xv[8];
      if (wave0 < nif[s]) {          // `cur` was fetched one stage ago exactly when this holds
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = *reinterpret_cast<const f32x4*>(rec + in_off + min(4 * q + 16 * j, K - 4));
      }
#if MFM_LAUNCH_STAMP
      if (s == 2 && threadIdx.x < 64) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); LSTAMP(LST_KF, 27); }
#endif
      if constexpr (!PRE) fetch(sn, nxt);
      mark(L, 2 + 2 * s);
#if MFM_LAUNCH_STAMP
      if (s == 2 && threadIdx.x < 64) { LSTAMP(LST_KF, 28); asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); LSTAMP(LST_KF, 29); }
#endif
      if (wave0 < nif[s]) {
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const f32x4 w4 = cur.w[j], x4 = xv[j];
          float p = w4[0] * x4[0];
          p = fmaf(w4[1], x4[1], p); p = fmaf(w4[2], x4[2], p); p = fmaf(w4[3], x4[3], p);
          p = (4 * q + 16 * j < K) ? p : 0.0f;
          if (j & 1) a1 += p; else a0 += p;
        }
        float v = a0 + a1;
        v += dpp_quad<0xB1>(v);
        v += dpp_quad<0x4E>(v);
        v += cur.bias;
        const int ew = cur.e[3];
        if (((ew >> 26) & 1) && q == 0) {
          const int out_idx = ew & 0xFFFF;
          if ((ew >> 24) & 1) v = fmaxf(v, 0.0f);
          if ((ew >> 25) & 1) {
            const int o = (ew >> 16) & 0xFF;
            const LatOp& op = ops[o];
            const int n = out_idx - op.out_off;
            float mk = 1.0f;
            if (L.train && op.drop_p > 0.0f) {
              const uint64_t idx = ((uint64_t)o << 40) + (uint64_t)row * (uint64_t)op.N + (uint64_t)n;
              mk = (rng_uniform(L.seed + (L.tick ? *L.tick : 0ull), idx) < op.drop_p) ? 0.0f : 1.0f / (1.0f - op.drop_p);
            }
            v *= mk;
            rec[op.mask_off + n] = mk;
          }
          rec[out_idx] = v;
        }
      }
    }
#if MFM_LAUNCH_STAMP
    if (s == 2) LSTAMP(LST_KF, 30);
#endif
    lds_barrier();
    mark(L, 3 + 2 * s);
    LSTAMP(LST_KF, 17 + s);
  };
  if constexpr (PRE) {
    Slot sl[LAT_PRE_SLOTS];
#pragma unroll
    for (int i = 0; i < LAT_PRE_SLOTS; ++i) fetch(min(i, L.nstages - 1), sl[i]);
#pragma unroll
    for (int i = 0; i < LAT_PRE_STAGES; ++i) {
      if (i < L.nstages) stage(i, sl[i % LAT_PRE_SLOTS], sl[i % LAT_PRE_SLOTS]);
      if (i + LAT_PRE_SLOTS < LAT_PRE_STAGES) fetch(min(i + LAT_PRE_SLOTS, L.nstages - 1), sl[i % LAT_PRE_SLOTS]);   // unconditional
    }
  } else {
    Slot sa, sb;
    const int s0 = (mode == 3) ? min(L.tail_from, L.nstages) : 0, s1 = min(L.nstages, (mode == 2) ? stage_hi : L.nstages);
    if (s0 < s1) fetch(s0, sa);
    for (int s = s0; s < s1; s += 2) {
      stage(s, sa, sb);
      if (s + 1 < s1) stage(s + 1, sb, sa);
    }
  }
  const bool head_only = !PRE && mode == 2 && stage_hi < L.nstages;       // the tail block does the rest
  const bool tail_only = !PRE && mode == 3;

  // ---- losses (one partial per workgroup, one atomic each)
  const bool ych = all || ch == 3;                // the workgroup that holds the classifier's outputs
  if (!head_only) {
  float kld = 0.0f;
  if (e_haslv) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (!all && m != ch) continue;
      for (int j = tid; j < e_zn[m]; j += nt) {
        const float mu = rec[e_mu[m] + j], lv = rec[e_lv[m] + j];
        kld += 1.0f + lv - mu * mu - expf(lv);
      }
    }
  }
  float disc = 0.0f;
  if (L.y && ych) {
    if (e_kind == 0) {
      if (tid < e_od) disc += fabsf(rec[e_yoff + tid] - yv);
    } else if (tid == 0) {
      const float* z = rec + e_yoff;
      float mx = z[0];
      for (int o = 1; o < e_od; ++o) mx = fmaxf(mx, z[o]);
      float se = 0.0f;
      for (int o = 0; o < e_od; ++o) se += expf(z[o] - mx);
      disc += (logf(se) + mx) - z[ylab];
    }
  }
  // only the first two waves can hold non-zero partials (z_n, od <= 128)
  if (tid < 128) {
    kld = wave_sum_dpp(kld);
    disc = wave_sum_dpp(disc);
    if ((tid & 63) == 0) { red[0][tid >> 6] = kld; red[1][tid >> 6] = disc; }
  }
  lds_barrier();
  LSTAMP(LST_KF, 25);
  if (tid == 0 && e_losses) {
    if (e_haslv) atomicAdd(e_losses + 4, -0.5f * (red[0][0] + red[0][1]));
    if (L.y && ych) {
      const float inv = (e_kind == 0) ? 1.0f / ((float)e_B * (float)e_od) : 1.0f / (float)e_B;
      atomicAdd(e_losses + 0, (red[1][0] + red[1][1]) * inv);
    }
  }
  }      // !head_only
  // ---- outputs: plain stores, nothing in this kernel waits for them
  const int fy = e_fn[3];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    if (!e_dec[m] || tail_only) continue;
    const int hd = fy + e_fn[m];
    // decoder input [f_y | f_m]: the y chain owns the first part (for all three decoders), chain m the second
    const int j0 = (all || ch == 3) ? 0 : fy, j1 = (all || ch == m) ? hd : fy;
    for (int j = j0 + tid; j < j1; j += nt)
      e_dec[m][(int64_t)row * e_ld[m] + j] = (j < fy) ? rec[e_fo[3] + j] : rec[e_fo[m] + (j - fy)];
  }
  if (e_yout && ych && !head_only)
    for (int o = tid; o < e_od; o += nt) e_yout[(int64_t)row * e_od + o] = rec[e_yoff + o];
  if (e_rec) {
    // the saved record: this workgroup's range of it (the whole row, or its chain's contiguous segments)
    const int lo4 = all ? 0 : (L.ch_lo[ch] >> 2), hi4 = all ? (e_rs >> 2) : (L.ch_hi[ch] >> 2);
    f32x4* d4 = reinterpret_cast<f32x4*>(e_rec + (int64_t)row * e_rs);
    const f32x4* s4 = reinterpret_cast<const f32x4*>(rec);
    for (int idx = lo4 + tid; idx < hi4; idx += nt) d4[idx] = s4[idx];
  }
  mark(L, 20);
}

// `mode` (round 6, the launch clock: 3.1 us of stores and atomics between the last stage and the BPTT that only needs d h_T):
// 0 = the whole body; 1 = everything up to the last stage -- the gradient record stays in LDS behind this body's tables
// (latent_bwd_grd_floats: the BPTT takes d h_T from there); 2 = what nothing in the workgroup waits for: bias gradients, d h_T and
// the gradient record to memory.  The fold launch runs 2 between the BPTT's weight requests and their first use.
// 3 (round 6) = a HEAD block of the decoder BPTT launch: the part of the backward chain that does not wait for the decoders -- the
// discriminative and KLD seeds and the stages from LatentDev::tail_from up (classifier, logvar heads) -- whose gradient record
// goes to grd_out; the chain of the encoder launch then starts from that record (LatentDev::bwd_split), adds the decoders' seeds
// and walks the remaining stages: 2.4 us of stages and most of the seed section leave the launch whose BPTT the chain gates.
__host__ __device__ static inline int latent_bwd_grd_floats(int rec_size) { return MFM_LAT_MAXSTAGES * MFM_LAT_ROW_THREADS * 4 + rec_size; }
template <bool PRE>
__device__ __forceinline__ void latent_bwd_row_body(const LatentDev& L, const float* __restrict__ params, float* __restrict__ grads,
                                                    const int row, const int ch, float* lds, const int mode = 0) {
#if MFM_LAUNCH_STAMP
  const int LST_KB = mode == 3 ? 3 : 4;        // a head block is a block of the decoder BPTT launch
#endif
  __shared__ LatOp ops[MFM_LAT_MAXOPS];
  __shared__ int pfxN[MFM_LAT_MAXOPS];
  const int RS = L.rec_size;
  i32x4* tab = reinterpret_cast<i32x4*>(lds);                        // [MAXSTAGES][1024] items
  float* rec = lds + MFM_LAT_MAXSTAGES * MFM_LAT_ROW_THREADS * 4;
  float* grd = rec + RS;
  const int nch = L.nch;                           // 4: one modality chain of the row per workgroup (see the forward)
  const bool all = nch == 1;
  const int tid = threadIdx.x, nt = blockDim.x;
  if (mode != 2) {
  int nib[MFM_LAT_MAXSTAGES];
#pragma unroll
  for (int i = 0; i < MFM_LAT_MAXSTAGES; ++i) nib[i] = L.nitems_bwd_c[ch][i];
  // The seeds' global operands (targets, upstream gradient, the decoders' d h_init) of this thread's element are requested
  // HERE, with the record: the seed section below used to load them where it needs them -- up to seven dependent round
  // trips (one per `if (ptr) sum += ptr[...]`) in front of the stage walk, on the critical path of the fold launches.
  // Unconditional uses (LAT_KEEP) keep the compiler from sinking the loads back into the branches that consume them.
#define LAT_KEEP(x) asm volatile("" : "+v"(x))
  // Descriptor fields of the seed section are fetched NOW and pinned in scalar registers (round 6; the launch clock: 3.0 us
  // between "tables in LDS" and "seeds done" for a few hundred instructions -- every first touch of a kernel-argument line read
  // where it is used is a scalar-cache miss in front of the stage walk; the forward body has done the same since round 2).
#define PIN_S(x) asm volatile("" : "+s"(x))
  int s_fn[4], s_fo[4], s_mu[4], s_lv[4], s_zn[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    s_fn[m] = L.f_n[m]; s_fo[m] = L.f_off[m]; s_mu[m] = L.mu_off[m]; s_lv[m] = L.lv_off[m]; s_zn[m] = L.z_n[m];
    PIN_S(s_fn[m]); PIN_S(s_fo[m]); PIN_S(s_mu[m]); PIN_S(s_lv[m]); PIN_S(s_zn[m]);
  }
  const bool head = !PRE && mode == 3, rest = !PRE && mode != 3 && L.bwd_split != 0;
  int s_yoff = L.yhat_off, s_od = L.od, s_B = L.B, s_kind = L.loss_kind, s_haslv = L.has_logvar;
  float s_genw = head ? 0.0f : L.gen_w, s_discw = L.disc_w, s_regw = L.reg_w;
  PIN_S(s_yoff); PIN_S(s_od); PIN_S(s_B); PIN_S(s_kind); PIN_S(s_haslv); PIN_S(s_genw); PIN_S(s_discw); PIN_S(s_regw);
  const float* s_ddi[3]; int64_t s_dld[3];
#pragma unroll
  for (int m = 0; m < 3; ++m) { s_ddi[m] = L.d_dec_init[m]; s_dld[m] = L.dec_ld[m]; PIN_S(s_ddi[m]); PIN_S(s_dld[m]); }
  const void* s_y = L.y; const float* s_dyext = L.d_yhat_ext; float* s_dlo = L.disc_loss_out; const float* s_rwp = L.reg_w_ptr;
  PIN_S(s_y); PIN_S(s_dyext); PIN_S(s_dlo); PIN_S(s_rwp);
#undef PIN_S
  // (every load is unconditional -- an absent operand reads params[0] instead: a branch around a load makes the compiler wait
  //  for everything requested before it)
  const int fy_ = s_fn[3];
  float pre_fy[3], pre_fm[3], pre_y;
  long long pre_lab;
  {
    const bool gen = L.gen_w != 0.0f && !head;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const bool on = gen && L.d_dec_init[m] != nullptr;
      const float* dp = on ? L.d_dec_init[m] + (int64_t)row * L.dec_ld[m] : params;
      pre_fy[m] = dp[on ? min(tid, fy_ - 1) : 0];
      pre_fm[m] = dp[on ? fy_ + min(tid, L.f_n[m] - 1) : 0];
    }
    const bool ext = L.d_yhat_ext != nullptr;
    const bool l1 = !ext && L.y && L.loss_kind == 0;
    const bool ce = !ext && L.y && L.loss_kind != 0;
    const float* yp = ext ? L.d_yhat_ext : (l1 ? reinterpret_cast<const float*>(L.y) : params);
    pre_y = yp[(ext || l1) ? (int64_t)row * L.od + min(tid, L.od - 1) : 0];
    const long long* lp = ce ? reinterpret_cast<const long long*>(L.y) : reinterpret_cast<const long long*>(params);
    pre_lab = lp[ce ? row : 0];
  }
  {   // op table, item table and the saved record are requested together (one round trip)
    const int nw = L.nops * (int)(sizeof(LatOp) / 4);
    const int opv = reinterpret_cast<const int*>(L.ops)[min(tid, nw - 1)];
    const int n4 = RS >> 2;
    const f32x4* s4 = reinterpret_cast<const f32x4*>(L.rec + (int64_t)row * RS);
    f32x4* r4 = reinterpret_cast<f32x4*>(rec);
    f32x4* g4 = reinterpret_cast<f32x4*>(grd);
    const f32x4 rv = s4[min(tid, n4 - 1)];
    const f32x4* sd4 = L.grd_seed ? reinterpret_cast<const f32x4*>(L.grd_seed + (int64_t)row * RS) : nullptr;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    const float sw = sd4 ? (L.seed_w_ptr ? *L.seed_w_ptr : L.seed_w) : 0.0f;
    // (no seed record: the saved record is read again and weighted by 0 -- an unconditional load, see above)
    const f32x4 sraw = (sd4 ? sd4 : s4)[min(tid, n4 - 1)];
    const int pfv = L.ops[min(tid, L.nops - 1)].pfx_n;
    // this thread's items of all stages: requested here, stored below (load_items() in two halves, so that every load of the
    // prologue is in flight before the first wait)
    const i32x4* isrc = reinterpret_cast<const i32x4*>(L.items_bwd + (size_t)ch * (MFM_LAT_MAXSTAGES * MFM_LAT_ROW_THREADS * 4));
    i32x4 iv[MFM_LAT_MAXSTAGES];
#pragma unroll
    for (int st = 0; st < MFM_LAT_MAXSTAGES; ++st)        // (the last slot: the bias-gradient table, LatentDev::bias_tab)
      iv[st] = isrc[(st == MFM_LAT_MAXSTAGES - 1 ? st : min(st, L.nstages - 1)) * MFM_LAT_ROW_THREADS + tid];
    const f32x4 sv = sw * sraw;
#pragma unroll
    for (int st = 0; st < MFM_LAT_MAXSTAGES; ++st) tab[st * MFM_LAT_ROW_THREADS + tid] = iv[st];
    if (tid < nw) reinterpret_cast<int*>(ops)[tid] = opv;
    if (tid < L.nops) pfxN[tid] = pfv;
    if (tid < n4) { r4[tid] = rv; g4[tid] = sv; }
    for (int idx = tid + nt; idx < n4; idx += nt) { r4[idx] = s4[idx]; g4[idx] = sd4 ? sw * sd4[idx] : zero; }
  }
  // (the unconditional uses come AFTER the record's loads were issued: placed above them they would wait for the seeds first)
#pragma unroll
  for (int m = 0; m < 3; ++m) { LAT_KEEP(pre_fy[m]); LAT_KEEP(pre_fm[m]); }
  LAT_KEEP(pre_y);
  {
    int lo = (int)pre_lab, hi = (int)(pre_lab >> 32);
    LAT_KEEP(lo); LAT_KEEP(hi);
    pre_lab = ((long long)hi << 32) | (unsigned)lo;
  }
#undef LAT_KEEP
  lds_barrier();
  LSTAMP(LST_KB, 16);
  const int l = tid & 15;
  const int wave0 = tid & ~63;
  struct Slot { f32x4 w[8]; i32x4 e; };
  // (rows beyond the layer's N: requested outside the parameter buffer -- zeros, no cache access; see the forward)
  const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)params, 0, (int)min((int64_t)0x7fffffff, L.n_params * 4), 0x00020000);
  auto fetch = [&](int s, Slot& t) {
    t.e = tab[s * MFM_LAT_ROW_THREADS + tid];
    const int K = t.e[1] & 0xFF, N = (t.e[1] >> 8) & 0xFF;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int n = l + 16 * j;
      t.w[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, n < N ? (int)(((unsigned)t.e[0] + n * K) * 4u) : 0x7FFFFFF0, 0, 0));
    }
  };
  Slot sa, sb, sl[PRE ? LAT_PRE_SLOTS : 1];
  // PRE: walk position p = 0 .. 5 is stage nstages-1-p; slot p % 4; the first four positions are requested here
  if constexpr (PRE) {
#pragma unroll
    for (int p = 0; p < LAT_PRE_SLOTS; ++p) fetch(max(L.nstages - 1 - p, 0), sl[p]);
  } else {
    fetch((rest ? L.tail_from : L.nstages) - 1, sa);
  }
  // ---- seeds
  float dl = 0.0f;                  // this row's share of the discriminative loss (disc_loss_out)
  // (the element a thread's first pass needs is already in pre_*; later passes -- od / f sizes beyond the block -- load)
  if (rest) {
    // (the head block seeded y_hat and the KLD terms and walked the classifier / logvar stages)
  } else if (s_dyext) {
    for (int o = tid; o < s_od; o += nt) grd[s_yoff + o] = (o == tid) ? pre_y : s_dyext[(int64_t)row * s_od + o];
  } else if (s_y && (s_discw != 0.0f || s_dlo)) {
    if (s_kind == 0) {
      const float* y = reinterpret_cast<const float*>(s_y);
      const float inv = 1.0f / ((float)s_B * (float)s_od);
      const float sc = s_discw * inv;
      for (int o = tid; o < s_od; o += nt) {
        const float df = rec[s_yoff + o] - ((o == tid) ? pre_y : y[(int64_t)row * s_od + o]);
        grd[s_yoff + o] = (df > 0.0f) ? sc : ((df < 0.0f) ? -sc : 0.0f);
        dl += fabsf(df) * inv;
      }
    } else if (tid == 0) {
      const float sc = s_discw / (float)s_B;
      const float* z = rec + s_yoff;
      float mx = z[0];
      for (int o = 1; o < s_od; ++o) mx = fmaxf(mx, z[o]);
      float se = 0.0f;
      for (int o = 0; o < s_od; ++o) se += expf(z[o] - mx);
      const int lab = (int)pre_lab;
      for (int o = 0; o < s_od; ++o) grd[s_yoff + o] = sc * (expf(z[o] - mx) / se - (o == lab ? 1.0f : 0.0f));
      dl = ((logf(se) + mx) - z[lab]) / (float)s_B;
    }
    // (od <= 128: only the first two waves hold a share; the chain workgroups of a row all seed y_hat, one of them reports)
    if (s_dlo && (all || ch == 3) && tid < 128) {
      dl = wave_sum_dpp(dl);
      if ((tid & 63) == 0 && dl != 0.0f) atomicAdd(s_dlo, dl);
    }
  }
  if (s_genw != 0.0f) {
    const int fy = s_fn[3];
    for (int j = tid; j < fy; j += nt) {
      float sm = 0.0f;
#pragma unroll
      for (int m = 0; m < 3; ++m)
        if (s_ddi[m]) sm += (j == tid) ? pre_fy[m] : s_ddi[m][(int64_t)row * s_dld[m] + j];
      grd[s_fo[3] + j] = rest ? grd[s_fo[3] + j] + sm : sm;          // (rest: on top of the classifier's contribution)
    }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      if (!s_ddi[m]) continue;
      for (int j = tid; j < s_fn[m]; j += nt)
        grd[s_fo[m] + j] = (j == tid) ? pre_fm[m] : s_ddi[m][(int64_t)row * s_dld[m] + fy + j];
    }
    // the f segments come out of a relu layer (z -> f, second Linear): the record holds gradients wrt
    // PRE-activations throughout (what the bias / weight gradients need), so the seeds are masked here and
    // every later contribution is masked where it is accumulated (stage loop)
#pragma unroll
    for (int m = 0; m < 4; ++m)
      for (int j = tid; j < s_fn[m]; j += nt)
        if (!(rec[s_fo[m] + j] > 0.0f)) grd[s_fo[m] + j] = 0.0f;
  }
  const float reg_w = s_rwp ? *s_rwp : s_regw;
  if (!rest && s_haslv && (s_rwp || reg_w != 0.0f)) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
      for (int j = tid; j < s_zn[m]; j += nt) {
        const float mu = rec[s_mu[m] + j], lv = rec[s_lv[m] + j];
        grd[s_mu[m] + j] = reg_w * mu;
        grd[s_lv[m] + j] = reg_w * (-0.5f) * (1.0f - expf(lv));
      }
  }
  lds_barrier();
  mark(L, 24);
  LSTAMP(LST_KB, 17);

  auto stage = [&](int s, Slot& cur, Slot& nxt) {
    const int sn = max(s - 1, 0);
    mark(L, 25 + 3 * s);
    // ---- pass 2a: dX[k] += sum_n g[n] W[n][k]   (per-wave skip and unconditional loads as in the forward)
    if (wave0 < (PRE ? nib[s] : max(nib[s], nib[sn]))) {
      const int N = (cur.e[1] >> 8) & 0xFF;
      const int out_off = cur.e[2] & 0xFFFF, in_idx = (cur.e[2] >> 16) & 0xFFFF;
      float gv[8];
      if (wave0 < nib[s]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) gv[j] = grd[out_off + min(l + 16 * j, N - 1)];
      }
      if constexpr (!PRE) fetch(sn, nxt);                // stage 0 requests its own weights again
      if (wave0 < nib[s]) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float gj = (l + 16 * j < N) ? gv[j] : 0.0f;
          acc += gj * cur.w[j];
        }
        float out[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float v = acc[c];
          v += dpp_row<0xB1>(v); v += dpp_row<0x4E>(v);
          v += dpp_row<0x141>(v); v += dpp_row<0x140>(v);
          out[c] = v;
        }
        const float lo = (l & 1) ? out[1] : out[0], hi = (l & 1) ? out[3] : out[2];
        const float v = (l & 2) ? hi : lo;
        const int ew = cur.e[3];
        if ((ew & 1) && l < 4) {
          float f = 1.0f;
          if ((ew & 2) && !(rec[in_idx + l] > 0.0f)) f = 0.0f;          // producer's relu
          if (ew >> 2) f *= rec[(ew >> 2) - 1 + l];                     // producer's dropout mask (scaled)
          atomicAdd(&grd[in_idx + l], v * f);
        }
      }
    }
    mark(L, 26 + 3 * s);
    lds_barrier();
    mark(L, 27 + 3 * s);
    LSTAMP(LST_KB, 18 + s);
  };
  if constexpr (PRE) {
#pragma unroll
    for (int p = 0; p < LAT_PRE_STAGES; ++p) {
      const int st = L.nstages - 1 - p;
      if (st >= 0) stage(st, sl[p % LAT_PRE_SLOTS], sl[p % LAT_PRE_SLOTS]);
      if (p + LAT_PRE_SLOTS < LAT_PRE_STAGES) fetch(max(L.nstages - 1 - (p + LAT_PRE_SLOTS), 0), sl[p % LAT_PRE_SLOTS]);
    }
  } else {
    const int lo = head ? L.tail_from : 0;
    for (int s = (rest ? L.tail_from : L.nstages) - 1; s >= lo; s -= 2) {
      stage(s, sa, sb);
      if (s - 1 >= lo) stage(s - 1, sb, sa);
    }
  }
  if (head) {        // the gradient record so far (this chain's range) -> memory, for the encoder launch's chain
    const int lo4 = all ? 0 : (L.ch_lo[ch] >> 2), hi4 = all ? (RS >> 2) : (L.ch_hi[ch] >> 2);
    f32x4* d4 = reinterpret_cast<f32x4*>(L.grd_out + (int64_t)row * RS);
    const f32x4* s4 = reinterpret_cast<const f32x4*>(grd);
    for (int idx = lo4 + tid; idx < hi4; idx += nt) d4[idx] = s4[idx];
    return;
  }
  }      // mode != 2
  if (mode == 1) return;

  // ---- bias gradients of all layers in one go (the record keeps every pre-activation gradient).  Inside
  // the stage loop these atomics would sit between two weight prefetches in the in-order vmcnt queue, and
  // every wait for weights would also wait for them.  Weight gradients: grouped GEMM over the two records.
  if (L.bias_tab && L.bias_n <= nt) {
    const i32x4 be = tab[(MFM_LAT_MAXSTAGES - 1) * MFM_LAT_ROW_THREADS + min(tid, MFM_LAT_ROW_THREADS - 1)];
    if (tid < L.bias_n && be[0] >= 0) atomicAdd(grads + (unsigned)be[0], grd[be[1]]);
  } else
  for (int st = 0; st < L.nstages; ++st) {
    const int ob = L.stage_begin[st], oe = L.stage_begin[st + 1];
    const int totn = L.nitems_fwd[st] >> 2;
    for (int item = tid; item < totn; item += nt) {
      const int o = find_op(pfxN, ob, oe, item, 1);
      const LatOp& op = ops[o];
      if (!all && op.chain != ch) continue;         // another workgroup of this row holds that layer's gradients
      const int n = item - pfxN[o];
      atomicAdd(grads + op.b_off + n, grd[op.out_off + n]);
    }
  }
  for (int m = 0; m < 4; ++m) {
    if (!L.dh_last[m] || (!all && m != ch)) continue;
    for (int k = tid; k < L.enc_n[m]; k += nt) L.dh_last[m][(int64_t)row * L.dh_ld[m] + k] = grd[L.in_off[m] + k];
  }
  if (L.grd_out) {
    const int lo4 = all ? 0 : (L.ch_lo[ch] >> 2), hi4 = all ? (RS >> 2) : (L.ch_hi[ch] >> 2);
    f32x4* d4 = reinterpret_cast<f32x4*>(L.grd_out + (int64_t)row * RS);
    const f32x4* s4 = reinterpret_cast<const f32x4*>(grd);
    if (L.grd_agent) {        // read by weight-gradient role workgroups of the same launch (dw_role_dev.h): agent-scope stores
      float* d1 = L.grd_out + (int64_t)row * RS;
      for (int idx = 4 * lo4 + tid; idx < 4 * hi4; idx += nt) __hip_atomic_store(d1 + idx, grd[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      for (int idx = lo4 + tid; idx < hi4; idx += nt) d4[idx] = s4[idx];
    }
  }
}

}  // namespace mfm
