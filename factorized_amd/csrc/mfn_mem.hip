// Memory Fusion Network: the sequential part of MFN.forward (reference mfm_model.py:177-181) as one
// persistent kernel per direction.
//
//   both   = [attended_t , mem]                                   (:177)
//   gamma1 = sigmoid(gamma1_fc2(drop(relu(gamma1_fc1(both)))))     (:178)
//   gamma2 = sigmoid(gamma2_fc2(drop(relu(gamma2_fc1(both)))))     (:179)
//   mem    = gamma1 * mem + gamma2 * cHat_t                         (:180)
//
// Everything else in the MFN time loop is independent of `mem` (the three LSTMs, the attention and
// cHat depend only on the cell states) and is evaluated for all T at once by the sequence kernels and
// grouped GEMMs; gamma*_fc1 is split by columns, W = [W_att | W_mem], and its attended part (with the
// bias) also arrives precomputed for all t.  What is left is a chain of T tiny steps
//   u_n = relu(att_n[t] + W_mem,n mem) ; gamma_n = sigmoid(W_fc2,n drop(u_n) + b_n) ; mem update
// which the reference runs as ~25 launches per step.  Here ONE workgroup owns one batch row for all T
// steps (B workgroups: the same latency regime as the small-tile LSTM kernels), all four weight
// matrices stay in VGPRs (canonical 4 x 8192 floats over 512 threads = 64 registers), mem / u /
// d-vectors are exchanged through LDS, partial dot products are all-reduced with DPP, two LDS-only
// barriers per step.  The per-step global operands (att, cHat, saved activations) are prefetched one
// step ahead with unconditional loads.
//
// Backward (BPTT) mirrors it with the transposed weights in registers:
//   dz_n = dmem * {mem_{t-1}, cHat_t} * gamma_n (1 - gamma_n) ; da_n = W_fc2,n^T dz_n ;
//   du_n = da_n * relu'/dropout ; dmem = dmem * gamma1 + W_mem,1^T du_1 + W_mem,2^T du_2
// and leaves dz_n, du_n, dcHat for all t; the weight gradients are sums over (t, b) of outer products of
// saved tensors and are formed afterwards by the grouped GEMM (host side: mfm_model.py::_MemFn).
#include <stdlib.h>

#include <algorithm>

#include "internal.h"

namespace mfm {

namespace {

constexpr int MEM_MAXW = 32;      // weights per thread and matrix

template <int CTRL>
__device__ __forceinline__ float dpp_m(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}
// all-reduce over groups of Q adjacent lanes, Q in {1,2,4,8,16}
__device__ __forceinline__ float group_sum(float v, int Q) {
  if (Q >= 2) v += dpp_m<0xB1>(v);      // quad_perm [1,0,3,2]
  if (Q >= 4) v += dpp_m<0x4E>(v);      // quad_perm [2,3,0,1]
  if (Q >= 8) v += dpp_m<0x141>(v);     // row_half_mirror
  if (Q >= 16) v += dpp_m<0x140>(v);    // row_mirror
  return v;
}

// Stores of the time loops go through buffer instructions whose per-lane offset is pushed out of range for lanes
// that have nothing to store: no store sits under a branch, so the compiler can count the memory operations between
// a prefetch and its use (vmcnt(n) instead of vmcnt(0): with branches every step waited for its own stores to be
// acknowledged, ~1 us each -- forward 52.8 -> ~25 us per launch at T = 20).
constexpr int OOB = 0x7FFFFFF0;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mem_rsrc(const float* p, int64_t elems) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)(elems * 4), 0x00020000);
}
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, int off_bytes, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, off_bytes, 0, 0);
}

// A thread's resident weights: elements [k0, k0 + MEM_MAXW) of one weight row (zero beyond n).  Whole, 16-byte aligned
// slices arrive as 8 independent 16-byte loads; the first version's 32 clamped dword loads per slice made the forward's
// prologue address-rate bound (96 loads x 32 cache lines per wave instruction: 18 of the 40 us of a T = 20 launch,
// scripts/bench_mfn_mem.py).
__device__ __forceinline__ void load_wslice(float (&w)[32], const float* __restrict__ row, int k0, int n, bool act) {
  const float* src = row + k0;
  if (act && k0 + 32 <= n && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    f32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = reinterpret_cast<const f32x4*>(src)[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) { w[4 * i] = v[i][0]; w[4 * i + 1] = v[i][1]; w[4 * i + 2] = v[i][2]; w[4 * i + 3] = v[i][3]; }
  } else {
#pragma unroll
    for (int i = 0; i < 32; ++i) w[i] = (act && k0 + i < n) ? row[min(k0 + i, n - 1)] : 0.0f;
  }
}

// Matvec operand layout (round 2): lane q of a QA- / QB-lane group owns the CONTIGUOUS slice k in [32 q, 32 q + 32) of the
// reduction dimension, so its 32 vector elements arrive as 8 ds_read_b128 (the first version interleaved k = q + Q i:
// 96 ds_read_b32 per thread and step made the kernels LDS-instruction bound, ~2 us per step).  LDS vectors are padded to
// 32 Q floats and zero-filled once; weights beyond the true extent are zero.
struct MemDev {
  MfmMemDesc d;
  int QA, KA, QB, KB1, KB2;       // lanes per output / weights per lane of the two matvec shapes
  int MP, H1P, H2P;               // LDS extents: 32 QA, 32 QB, 32 QB
  int64_t ldw;                    // row stride of the memory-column weight blocks
  MfnHeadsDev hd;                 // heads on [h_T | mem_T] folded into the launch (internal.h); hd.on == 0: not used
};

// ------------------------------------------------------------------------------------- forward
// MAXT = 512 gives the compiler 256 registers per thread: the 96 resident weights of the canonical shape
// (M=64, H=128: 512 threads) then stay in VGPRs; the 1024-thread build (128 registers) spills them.
template <int MAXT>
__global__ __launch_bounds__(MAXT) void mfn_mem_fwd_kernel(const MemDev P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const MfmMemDesc& d = P.d;
  const int T = d.T, B = d.B, M = d.M, H1 = d.H1, H2 = d.H2;
  const int QA = P.QA, QB = P.QB;
  const int MP = P.MP, H1P = P.H1P;
  float* memb = lds;                 // [2][MP]
  float* ab = lds + 2 * MP;          // [H1P + H2P] activations of the step
  const int row = blockIdx.x;
  const int tid = threadIdx.x;

  // ---- phase-A role: (net, j, q): u_net[j] = sum_k Wm_net[j][k] mem[k], k = q + QA i
  const int nA = (H1 + H2) * QA;
  const bool actA = tid < nA;
  const int ja = min(tid, nA - 1) / QA, qa = tid % QA;
  const int netA = ja >= H1;
  const int jn = netA ? ja - H1 : ja;
  const float* wm = netA ? d.w2m : d.w1m;
  float wA[MEM_MAXW];
  load_wslice(wA, wm + (int64_t)jn * P.ldw, MEM_MAXW * qa, M, actA);
  float* abuf = netA ? d.a2 : d.a1;
  const int Hn = netA ? H2 : H1;
  const float pA = netA ? d.p2 : d.p1;
  const float keepA = (pA < 1.0f) ? 1.0f / (1.0f - pA) : 0.0f;
  // ---- phase-B role: (m, q): z_n[m] = sum_k Wb_n[m][k] a_n[k], both nets
  const int nB = M * QB;
  const bool actB = tid < nB;
  const int mb = min(tid, nB - 1) / QB, qb = tid % QB;
  float wB1[MEM_MAXW], wB2[MEM_MAXW];
  load_wslice(wB1, d.w1b + (int64_t)mb * H1, MEM_MAXW * qb, H1, actB);
  load_wslice(wB2, d.w2b + (int64_t)mb * H2, MEM_MAXW * qb, H2, actB);
  const float bb1 = d.b1b[mb], bb2 = d.b2b[mb];

  const uint64_t seed = d.seed + (d.seed_dev ? *d.seed_dev : 0ull);
  for (int i = tid; i < 2 * MP + H1P + P.H2P; i += blockDim.x) lds[i] = 0.0f;
  float memr = 0.0f;                              // mem[mb], carried by lane q == 0 of the phase-B group
  const int64_t arow = ((int64_t)row) * Hn + jn;  // + t * B * Hn
  const int64_t mrow = ((int64_t)row) * M + mb;   // + t * B * M
  float att_n = abuf[arow];
  float ch_n = d.chat[mrow];
  lds_barrier();
  const int64_t TBl = (int64_t)T * B;
  const __amdgpu_buffer_rsrc_t ra1 = mem_rsrc(d.a1, TBl * H1), ra2 = mem_rsrc(d.a2, TBl * H2);
  const __amdgpu_buffer_rsrc_t rg1 = mem_rsrc(d.gam1, TBl * M), rg2 = mem_rsrc(d.gam2, TBl * M), rmm = mem_rsrc(d.mems, TBl * M);
  const bool stA = actA && qa == 0, stB = actB && qb == 0;

  int cur = 0;
  for (int t = 0; t < T; ++t) {
    const float att = att_n, ch = ch_n;
    const int tn = min(t + 1, T - 1);
    att_n = abuf[(int64_t)tn * B * Hn + arow];     // unconditional prefetch (clamped at the tail)
    ch_n = d.chat[(int64_t)tn * B * M + mrow];
    // ---- phase A
    {
      const f32x4* mp = reinterpret_cast<const f32x4*>(memb + cur * MP + MEM_MAXW * qa);
      float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
      for (int i = 0; i < MEM_MAXW / 4; ++i) {
        const f32x4 v = mp[i];
        s0 = fmaf(wA[4 * i], v[0], s0); s1 = fmaf(wA[4 * i + 1], v[1], s1);
        s0 = fmaf(wA[4 * i + 2], v[2], s0); s1 = fmaf(wA[4 * i + 3], v[3], s1);
      }
      float u = group_sum(s0 + s1, QA) + att;
      u = fmaxf(u, 0.0f);
      if (d.train && pA > 0.0f) {
        const uint64_t idx = ((uint64_t)(netA + 1) << 56) + ((uint64_t)t << 40) + (uint64_t)row * (uint64_t)Hn + (uint64_t)jn;
        u = (rng_uniform(seed, idx) < pA) ? 0.0f : u * keepA;
      }
      if (stA) ab[netA ? H1P + jn : jn] = u;
      {   // saved for the backward, in place over the input
        const int off = (int)(((int64_t)t * B * Hn + arow) * 4);
        bstore(ra1, (stA && !netA) ? off : OOB, u);
        bstore(ra2, (stA && netA) ? off : OOB, u);
      }
    }
    lds_barrier();
    // ---- phase B + memory update
    {
      const f32x4* a1p = reinterpret_cast<const f32x4*>(ab + MEM_MAXW * qb);
      const f32x4* a2p = reinterpret_cast<const f32x4*>(ab + H1P + MEM_MAXW * qb);
      float z1 = 0.0f, z2 = 0.0f;
#pragma unroll
      for (int i = 0; i < MEM_MAXW / 4; ++i) {
        const f32x4 v1 = a1p[i], v2 = a2p[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) { z1 = fmaf(wB1[4 * i + e], v1[e], z1); z2 = fmaf(wB2[4 * i + e], v2[e], z2); }
      }
      z1 = group_sum(z1, QB) + bb1;
      z2 = group_sum(z2, QB) + bb2;
      const float g1 = act_sigmoid(z1), g2 = act_sigmoid(z2);
      memr = g1 * memr + g2 * ch;
      if (stB) memb[(cur ^ 1) * MP + mb] = memr;
      {
        const int off = stB ? (int)(((int64_t)t * B * M + mrow) * 4) : OOB;
        bstore(rg1, off, g1); bstore(rg2, off, g2); bstore(rmm, off, memr);
      }
    }
    lds_barrier();
    cur ^= 1;
  }
  if (actB && qb == 0 && d.mem_out) d.mem_out[mrow] = memr;
  // ---- heads on mfn_last = [h_l, h_a, h_v](T-1) | mem_T: 8 lanes per output, K = tot + M.  The input vector is staged in
  // LDS first (over the recurrence's buffers, which nobody reads any more), so the dot products run without the
  // per-element segment branches and their loads go out back to back
  if (P.hd.on) {
    const MfnHeadsDev& H = P.hd;
    const int KT = H.tot + M, nout = H.nheads * H.zy;
    const int q8 = tid & 7;
    const int n0 = H.seg_n[0], n1 = n0 + H.seg_n[1];
    const bool staged = KT <= 2 * MP + H1P + P.H2P;          // (wave-uniform)
    float* vec = lds;
    if (staged) {
      for (int k = tid; k < H.tot; k += blockDim.x) {
        float v;
        if (k < n0) v = H.seg[0][(int64_t)row * H.seg_ld[0] + k];
        else if (k < n1) v = H.seg[1][(int64_t)row * H.seg_ld[1] + (k - n0)];
        else v = H.seg[2][(int64_t)row * H.seg_ld[2] + (k - n1)];
        vec[k] = v;
      }
      if (stB) vec[H.tot + mb] = memr;
      __syncthreads();
    }
    for (int o = tid >> 3; o < nout; o += (int)blockDim.x >> 3) {
      const int hd = o / H.zy, n = o - hd * H.zy;
      const float* wr = H.w[hd] + (int64_t)n * KT;
      float s = 0.0f;
      if (staged) {
        float s1 = 0.0f;
        int k = q8;
        for (; k + 56 < KT; k += 64) {
          float wv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) wv[u] = wr[k + 8 * u];
#pragma unroll
          for (int u = 0; u < 8; u += 2) { s = fmaf(wv[u], vec[k + 8 * u], s); s1 = fmaf(wv[u + 1], vec[k + 8 * u + 8], s1); }
        }
        for (; k < KT; k += 8) s = fmaf(wr[k], vec[k], s);
        s += s1;
      } else {
#pragma unroll 8
        for (int k = q8; k < KT; k += 8) {
          float v;
          if (k < n0) v = H.seg[0][(int64_t)row * H.seg_ld[0] + k];
          else if (k < n1) v = H.seg[1][(int64_t)row * H.seg_ld[1] + (k - n0)];
          else if (k < H.tot) v = H.seg[2][(int64_t)row * H.seg_ld[2] + (k - n1)];
          else v = memb[cur * MP + (k - H.tot)];
          s = fmaf(wr[k], v, s);
        }
      }
      s = group_sum(s, 8);
      if (q8 == 0) H.zyin[(int64_t)row * H.nzy + hd * H.zy + n] = s + H.b[hd][n];
    }
  }
}

// ------------------------------------------------------------------------------------- backward
template <int MAXT>
__global__ __launch_bounds__(MAXT) void mfn_mem_bwd_kernel(const MemDev P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const MfmMemDesc& d = P.d;
  const int T = d.T, B = d.B, M = d.M, H1 = d.H1, H2 = d.H2;
  const int QA = P.QA, QB = P.QB;
  const int MP = P.MP, H1P = P.H1P;
  float* dzb = lds;                  // [2][MP]  dz of net 1 | net 2
  float* dub = lds + 2 * MP;         // [H1P + H2P]
  for (int i = threadIdx.x; i < 2 * MP + H1P + P.H2P; i += blockDim.x) lds[i] = 0.0f;
  lds_barrier();
  const int row = blockIdx.x;
  const int tid = threadIdx.x;

  // ---- role A (transposed fc2): da_net[j] = sum_m Wb_net[m][j] dz_net[m], m = q + QA i
  const int nA = (H1 + H2) * QA;
  const bool actA = tid < nA;
  const int ja = min(tid, nA - 1) / QA, qa = tid % QA;
  const int netA = ja >= H1;
  const int jn = netA ? ja - H1 : ja;
  const int Hn = netA ? H2 : H1;
  const float* wb = netA ? d.w2b : d.w1b;
  float wA[MEM_MAXW];
#pragma unroll
  for (int i = 0; i < MEM_MAXW; ++i) {
    const int m = MEM_MAXW * qa + i;
    wA[i] = (actA && m < M) ? wb[(int64_t)min(m, M - 1) * Hn + jn] : 0.0f;
  }
  const float* abuf = netA ? d.a2 : d.a1;
  const float pA = netA ? d.p2 : d.p1;
  const float keepA = (d.train && pA > 0.0f) ? ((pA < 1.0f) ? 1.0f / (1.0f - pA) : 0.0f) : 1.0f;
  // ---- role B (transposed memory columns): dmem[m] += sum_j Wm_n[j][m] du_n[j], both nets
  const int nB = M * QB;
  const bool actB = tid < nB;
  const int mb = min(tid, nB - 1) / QB, qb = tid % QB;
  float wB1[MEM_MAXW], wB2[MEM_MAXW];
#pragma unroll
  for (int i = 0; i < MEM_MAXW; ++i) {
    const int j = MEM_MAXW * qb + i;
    wB1[i] = (actB && j < H1) ? d.w1m[(int64_t)min(j, H1 - 1) * P.ldw + mb] : 0.0f;
    wB2[i] = (actB && j < H2) ? d.w2m[(int64_t)min(j, H2 - 1) * P.ldw + mb] : 0.0f;
  }
  const int64_t arow = ((int64_t)row) * Hn + jn;
  const int64_t mrow = ((int64_t)row) * M + mb;
  float dmem = d.dmem_out ? d.dmem_out[mrow] : 0.0f;     // dL/d mem_T
  if (P.hd.on) {
    // through the heads: d mem_T[m] = sum_{head, n} dz[n] W[n][tot + m] (every lane of the group of m), and
    // d h_T[j] = sum dz[n] W[n][j] for the three LSTMs' last hidden states (one output per thread)
    const MfnHeadsDev& H = P.hd;
    const int KT = H.tot + M;
    float acc = 0.0f;
    for (int hd = 0; hd < H.nheads; ++hd) {
#pragma unroll 16
      for (int n = 0; n < H.zy; ++n) acc = fmaf(H.dz[(int64_t)row * H.nzy + hd * H.zy + n], H.w[hd][(int64_t)n * KT + H.tot + mb], acc);
    }
    dmem = acc;
    for (int j = tid; j < H.tot; j += blockDim.x) {
      float a = 0.0f;
      for (int hd = 0; hd < H.nheads; ++hd) {
#pragma unroll 16
        for (int n = 0; n < H.zy; ++n) a = fmaf(H.dz[(int64_t)row * H.nzy + hd * H.zy + n], H.w[hd][(int64_t)n * KT + j], a);
      }
      H.d_hT[(int64_t)row * H.tot + j] = a;
    }
  }
  // saved operands of step T-1
  int64_t o = (int64_t)(T - 1) * B * M + mrow;
  float g1_n = d.gam1[o], g2_n = d.gam2[o], ch_n = d.chat[o];
  float mp_n = (T > 1) ? d.mems[o - (int64_t)B * M] : 0.0f;
  float a_n = abuf[(int64_t)(T - 1) * B * Hn + arow];
  const int64_t TBl = (int64_t)T * B;
  const __amdgpu_buffer_rsrc_t rg1 = mem_rsrc(d.gam1, TBl * M), rg2 = mem_rsrc(d.gam2, TBl * M), rdc = mem_rsrc(d.dchat, TBl * M);
  const __amdgpu_buffer_rsrc_t ru1 = mem_rsrc(d.du1, TBl * H1), ru2 = mem_rsrc(d.du2, TBl * H2);
  const bool stA = actA && qa == 0, stB = actB && qb == 0;
  const bool pre_tanh = d.dchat_pre_tanh != 0;

  for (int t = T - 1; t >= 0; --t) {
    const float g1 = g1_n, g2 = g2_n, ch = ch_n, mprev = mp_n, av = a_n;
    {   // unconditional prefetch of step t-1 (clamped at t = 0; mem_{-1} = 0 by the multiply)
      const int tp = max(t - 1, 0);
      const int64_t op = (int64_t)tp * B * M + mrow;
      g1_n = d.gam1[op]; g2_n = d.gam2[op]; ch_n = d.chat[op];
      mp_n = d.mems[(int64_t)max(tp - 1, 0) * B * M + mrow] * (tp > 0 ? 1.0f : 0.0f);
      a_n = abuf[(int64_t)tp * B * Hn + arow];
    }
    // ---- phase 1: gate gradients
    const float dz1 = dmem * mprev * g1 * (1.0f - g1);
    const float dz2 = dmem * ch * g2 * (1.0f - g2);
    if (stB) { dzb[mb] = dz1; dzb[MP + mb] = dz2; }
    {   // in place: the weight-gradient GEMMs read dz from gam1 / gam2
      const int off = stB ? (int)(((int64_t)t * B * M + mrow) * 4) : OOB;
      bstore(rg1, off, dz1); bstore(rg2, off, dz2);
      bstore(rdc, off, pre_tanh ? dmem * g2 * (1.0f - ch * ch) : dmem * g2);
    }
    const float dmem_direct = dmem * g1;
    lds_barrier();
    // ---- phase 2: through gamma*_fc2 and the relu / dropout
    {
      const f32x4* zp = reinterpret_cast<const f32x4*>(dzb + netA * MP + MEM_MAXW * qa);
      float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
      for (int i = 0; i < MEM_MAXW / 4; ++i) {
        const f32x4 v = zp[i];
        s0 = fmaf(wA[4 * i], v[0], s0); s1 = fmaf(wA[4 * i + 1], v[1], s1);
        s0 = fmaf(wA[4 * i + 2], v[2], s0); s1 = fmaf(wA[4 * i + 3], v[3], s1);
      }
      const float da = group_sum(s0 + s1, QA);
      const float du = (av > 0.0f) ? da * keepA : 0.0f;
      if (stA) dub[netA ? H1P + jn : jn] = du;
      {
        const int off = (int)(((int64_t)t * B * Hn + arow) * 4);
        bstore(ru1, (stA && !netA) ? off : OOB, du);
        bstore(ru2, (stA && netA) ? off : OOB, du);
      }
    }
    lds_barrier();
    // ---- phase 3: into the memory
    {
      const f32x4* u1p = reinterpret_cast<const f32x4*>(dub + MEM_MAXW * qb);
      const f32x4* u2p = reinterpret_cast<const f32x4*>(dub + H1P + MEM_MAXW * qb);
      float s = 0.0f, s2 = 0.0f;
#pragma unroll
      for (int i = 0; i < MEM_MAXW / 4; ++i) {
        const f32x4 v1 = u1p[i], v2 = u2p[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) { s = fmaf(wB1[4 * i + e], v1[e], s); s2 = fmaf(wB2[4 * i + e], v2[e], s2); }
      }
      dmem = dmem_direct + group_sum(s + s2, QB);
    }
  }
}

int pow2_ge(int x) { int p = 1; while (p < x) p <<= 1; return p; }

int mem_setup(const MfmMemDesc* desc, MemDev& P, int& threads, size_t& lds) {
  const MfmMemDesc& d = *desc;
  MFM_REQUIRE(d.T >= 1 && d.B >= 1 && d.M >= 1 && d.H1 >= 1 && d.H2 >= 1, "mfn_mem: bad dims T=%d B=%d M=%d H=%d/%d", d.T, d.B, d.M, d.H1, d.H2);
  MFM_REQUIRE(d.a1 && d.a2 && d.chat && d.w1m && d.w2m && d.w1b && d.b1b && d.w2b && d.b2b && d.gam1 && d.gam2 && d.mems,
              "mfn_mem: null operand");
  {
    const int64_t widest = std::max<int64_t>(d.M, std::max(d.H1, d.H2));
    MFM_REQUIRE((int64_t)d.T * d.B * widest * 4 < ((int64_t)1 << 31), "mfn_mem: T*B*width %lld exceeds the 2 GB buffer-addressing range",
                (long long)((int64_t)d.T * d.B * widest));
  }
  P.d = d;
  P.ldw = d.ld_wm > 0 ? d.ld_wm : d.M;
  P.QA = pow2_ge(cdiv(d.M, MEM_MAXW));
  P.KA = cdiv(d.M, P.QA);
  const int hmax = d.H1 > d.H2 ? d.H1 : d.H2;
  P.QB = pow2_ge(cdiv(hmax, MEM_MAXW));
  P.KB1 = cdiv(d.H1, P.QB);
  P.KB2 = cdiv(d.H2, P.QB);
  const int nA = (d.H1 + d.H2) * P.QA, nB = d.M * P.QB;
  if (P.QA > 16 || P.QB > 16 || nA > 1024 || nB > 1024) {
    set_error("mfn_mem: memory %d / gate hidden %d,%d do not fit the register-resident recurrence", d.M, d.H1, d.H2);
    return MFM_ERR_UNSUPPORTED;
  }
  P.MP = MEM_MAXW * P.QA; P.H1P = MEM_MAXW * P.QB; P.H2P = MEM_MAXW * P.QB;
  threads = round_up(nA > nB ? nA : nB, 64);
  lds = (size_t)(2 * P.MP + P.H1P + P.H2P) * sizeof(float);
  return MFM_OK;
}

}  // namespace

}  // namespace mfm

using namespace mfm;

int mfm::mfn_mem_fwd_launch(const MfmMemDesc* desc, const MfnHeadsDev* heads, hipStream_t stream) {
  if (!desc) { set_error("mfm_mfn_mem_fwd: null descriptor"); return MFM_ERR_ARG; }
  MemDev P;
  int threads = 0; size_t lds = 0;
  int rc = mem_setup(desc, P, threads, lds);
  if (rc != MFM_OK) return rc;
  memset(&P.hd, 0, sizeof(P.hd));
  if (heads && heads->on) {
    MFM_REQUIRE(heads->nheads >= 1 && heads->nheads <= 2 && heads->zyin && heads->w[0] && heads->b[0] && threads >= 64,
                "mfn_mem: heads descriptor");
    P.hd = *heads;
  }
  if (threads <= 512) MFM_LAUNCH_TIMED(mfn_mem_fwd_kernel<512>, dim3(desc->B), dim3(threads), lds, (hipStream_t)stream, P);
  else MFM_LAUNCH_TIMED(mfn_mem_fwd_kernel<1024>, dim3(desc->B), dim3(threads), lds, (hipStream_t)stream, P);
  MFM_LAUNCH_CHECK("mfn_mem_fwd_kernel");
  return MFM_OK;
}

extern "C" int mfm_mfn_mem_fwd(const MfmMemDesc* desc, void* stream) {
  return mfm::mfn_mem_fwd_launch(desc, nullptr, (hipStream_t)stream);
}

int mfm::mfn_mem_bwd_launch(const MfmMemDesc* desc, const MfnHeadsDev* heads, hipStream_t stream) {
  if (!desc) { set_error("mfm_mfn_mem_bwd: null descriptor"); return MFM_ERR_ARG; }
  MemDev P;
  int threads = 0; size_t lds = 0;
  int rc = mem_setup(desc, P, threads, lds);
  if (rc != MFM_OK) return rc;
  MFM_REQUIRE(desc->du1 && desc->du2 && desc->dchat, "mfm_mfn_mem_bwd: null gradient output");
  memset(&P.hd, 0, sizeof(P.hd));
  if (heads && heads->on) {
    MFM_REQUIRE(heads->nheads >= 1 && heads->nheads <= 2 && heads->dz && heads->d_hT && heads->w[0], "mfn_mem: heads descriptor (backward)");
    P.hd = *heads;
  }
  if (threads <= 512) MFM_LAUNCH_TIMED(mfn_mem_bwd_kernel<512>, dim3(desc->B), dim3(threads), lds, (hipStream_t)stream, P);
  else MFM_LAUNCH_TIMED(mfn_mem_bwd_kernel<1024>, dim3(desc->B), dim3(threads), lds, (hipStream_t)stream, P);
  MFM_LAUNCH_CHECK("mfn_mem_bwd_kernel");
  return MFM_OK;
}

extern "C" int mfm_mfn_mem_bwd(const MfmMemDesc* desc, void* stream) {
  return mfm::mfn_mem_bwd_launch(desc, nullptr, (hipStream_t)stream);
}
