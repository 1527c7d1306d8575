// Input projections as ROLE workgroups of the encoder recurrence launch (round 3, MFM_KL_EF at B <= 32).
//
// At the reference's batch size the four encoder recurrences occupy 4 B <= 128 of the 256 CUs for ~34 us and the
// projection GEMM in front of them (x_t W_ih^T + b_ih + b_hh for all t, mfm_model.py:56 via nn.LSTMCell) is a launch of
// its own: ~16 us of cold misses for 0.26 GFLOP, plus the launch boundary.  Here the idle CUs produce the projections
// INSIDE the recurrence launch, time step by time step, ahead of the recurrence workgroups that consume them:
//   * role workgroup r (blockIdx < n_role, dispatched first) owns one 32-column block of one encoder's 4 Hp gate columns
//     for good -- its W_ih rows stay in LDS -- and walks the time steps t = r / 32, + n_role / 32, ...: the B x k slice of
//     x_t by 16-byte loads (the next step's slice is requested before the current product starts), 16 waves = 4
//     16x16 fragments x 4 k-parts on v_mfma_f32_16x16x4_f32, partial tiles joined through LDS, bias added, stored with
//     agent-scope stores; when every store of the tile is acknowledged, flags[e][t][block] <- epoch;
//   * the recurrence workgroup of (encoder e, row b) polls the flags of (e, t + 2) one step before it fetches that
//     step's projections (lstm_seq_small.hip, small_fwd_body<.., FLG = true>).
// No deadlock: producers never wait, and every XCD dispatches its workgroups in block order, so whatever a consumer
// waits for is resident or finished.  Flags are epoch stamps (the plan's call counter; the workspace starts zeroed):
// nothing to reset, nothing accumulates.  The launch's zero spans (gradient buffer, dH) ride on the role workgroups too.
#pragma once
#include "internal.h"
#include "lstm_seq_dev.h"
#include "lstamp.h"

namespace mfm {

constexpr int PROJ_ROLE_CB = 32;          // gate columns per block == time-step rows per item (B <= 32)
constexpr int PROJ_ROLE_SLOTS = 32;       // column-block slots per time-step group (4 Hp / 32 summed over the encoders, padded)
constexpr int PROJ_ROLE_MAXG = 4;         // 16-byte groups per thread and operand slice (k <= 512: the YouTube-shape early-fusion encoder has k = 410)
constexpr int PROJ_ROLE_FLAGS = 16;       // flag words per (encoder, time step)
constexpr int PROJ_ROLE_PW = 4 * 96 + 4;  // floats per wave in the partial-tile buffer ([reg][96] + skew)

struct ProjRoleEnc { const float* w; const float* b_ih; const float* b_hh; int k_off, k, cb_begin, ncb; };
constexpr int PROJ_ROLE_WT = 7;
// What a consumer does when its producer does not show up (shared by the projection and weight-gradient hand-overs): after
// `timeout` ticks of the 100 MHz wall clock it gives up, ORs `bit` into the plan's sticky STATUS word (the host reads it at its
// next sync, raises and switches the plan to separate launches; every backward of a plan whose status is non-zero poisons the
// gradient guard, so the optimizer skips the step: include/mfm_hip.h, plan options) and stores a NaN into `poison`.
// `host`: two words of host-coherent memory the plan owns (mfm_plan_host_status): the host sees a failure at its next glance at
// plain memory, without a copy or a synchronisation (word 0: projections, word 1: weight gradients).
struct HoCtl { unsigned* status; unsigned* host; float* poison; long long timeout; unsigned bit; };
__device__ __forceinline__ void ho_give_up(const HoCtl& c) {
  if ((threadIdx.x & 63) == 0) {
    if (c.status) __hip_atomic_fetch_or(c.status, c.bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (c.host) __hip_atomic_store(c.host + (c.bit >> 1), c.bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (c.poison) __hip_atomic_store(c.poison, __builtin_nanf(""), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// epoch of a launch = host part (a function of the plan's call counter) + a device word that only captured hipGraphs advance
// (one tick node per replay: plan.hip), never 0 (what a fresh workspace holds)
__device__ __forceinline__ unsigned ho_epoch(unsigned base, const unsigned* tick) {
  const unsigned e = base + (tick ? *tick : 0u);
  return e ? e : 0x5bd1e995u;
}

struct ProjRole {
  const float* x; int ldx; int x_rows;    // x[T * B, ldx]
  int n_role, groups, kstride;            // groups = n_role / 32
  unsigned* flags; unsigned epoch;        // flags[(e * T + t) * 16 + block]; epoch: host part (ho_epoch)
  const unsigned* tick;                   // device part of the epoch (low word of the plan's replay counter)
  HoCtl ctl;                              // consumer side: time-out, status word, poison
  int fault;                              // fault injection (tests): role workgroup 0 does not raise its first flag
  int lat_pre;                            // the rows' latent chains preload their tables in front of the time loop (lstm_seq_small.hip)
  int lat_tail;                           // ... and stop behind the z -> f MLPs: tail blocks of the decoder launch finish them
  float* loss_ptr; int loss_n;            // loss slots: cleared with agent-scope stores before any flag of t = 0 is raised
  int bf16;                               // bf16 plans: x and W_ih rounded to bf16 (RNE) on the way into LDS, fp32 accumulation
  ProjRoleEnc e[4];
  ZeroSpans zs;                           // cleared with plain stores (read by later launches only)
  WtImgItem wt[PROJ_ROLE_WT]; int n_wt;   // transposed-weight images to produce (training steps)
  WtImgItem wf[6]; int n_wf;              // forward-order images of the decoders' weights: W_ih (step 0), W_ih + W_hh (lstm_seq_dev.h)
};

// LDS row stride of the operand images: k padded to 16, then to an odd number of 16-byte groups (the 16 rows of a fragment
// read land on 64 distinct banks)
static inline int proj_role_kstride(int k) { int s4 = ((k + 15) / 16 * 16) / 4; if ((s4 & 1) == 0) ++s4; return 4 * s4; }
bool seq_small_foldproj_supported(int T, int B, const int* h, const int* k, int n_enc);
int seq_small_foldproj_launch(SeqLaunch& L, const LatentDev& LD, ProjRole& PR, const float* params, hipStream_t stream);
int seq_foldproj_launch(const MfmSeqDesc* descs, int count, int T, int B, const LatentDev& lat, const float* params, ProjRole& pr,
                        hipStream_t stream);

__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_agent_u(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// consumer side: true when every block flag of one (encoder, time step) carries this launch's epoch (wave-uniform)
__device__ __forceinline__ bool proj_flags_ready(unsigned v, unsigned epoch, int ncb) {
  const int lane = threadIdx.x & 63;
  return __builtin_amdgcn_ballot_w64(lane < ncb && v != epoch) == 0ull;
}
__device__ __forceinline__ unsigned proj_flags_load(const unsigned* f, int ncb) {
  const int lane = threadIdx.x & 63;
  return ld_agent_u(f + (lane < ncb ? lane : 0));
}
// spin until ready; gives up after ctl.timeout (default ~50 ms) so that a broken or blocked producer becomes a reported,
// survivable failure (ho_give_up), not a hung GPU and not a plausible-looking result
__device__ __forceinline__ void proj_flags_wait(const unsigned* f, unsigned v, unsigned epoch, int ncb, const HoCtl& ctl) {
  if (proj_flags_ready(v, epoch, ncb)) return;
  const long long t0 = wall_clock64();
  do {
    __builtin_amdgcn_s_sleep(2);
    v = proj_flags_load(f, ncb);
    if (wall_clock64() - t0 >= ctl.timeout) { ho_give_up(ctl); return; }
  } while (!proj_flags_ready(v, epoch, ncb));
}

__device__ __forceinline__ void proj_role_body(const SeqLaunch& L, const ProjRole& PR, const unsigned epoch, float* lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = blockIdx.x;
  const int slot = r & (PROJ_ROLE_SLOTS - 1), grp = r >> 5;
  const int T = L.T, B = L.B;
  int e = -1;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (slot >= PR.e[i].cb_begin && slot < PR.e[i].cb_begin + PR.e[i].ncb) e = i;

  if (e >= 0 && grp < T) {
    const ProjRoleEnc& E = PR.e[e];
    const SeqDev& d = L.d[e];
    const int K = E.k, h = d.h, Hp = d.Hp;
    const int cbl = slot - E.cb_begin;
    const int col0 = cbl * PROJ_ROLE_CB;
    const int ks = PR.kstride;
    const int G = ((K + 15) >> 4) << 2;            // 16-byte groups per operand row (k padded to 16 with zeros)
    float* Ws = lds;                               // [32][ks]
    float* Xs = lds + PROJ_ROLE_CB * ks;           // [32][ks]: ONE image -- the next slice waits in registers and is parked behind
                                                   // the barrier that ends the product's reads (round 4: k = 410 fits the LDS)
    float* Ps = Xs + PROJ_ROLE_CB * ks;            // [16 waves][PROJ_ROLE_PW]

    const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)E.w, 0, 4 * h * K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)PR.x, 0, (PR.x_rows * PR.ldx) * 4, 0x00020000);

    // slice element groups of this thread: i -> (row of the 32-row slice, 16-byte group of its k extent)
    int lrow[PROJ_ROLE_MAXG], lgr[PROJ_ROLE_MAXG];
#pragma unroll
    for (int j = 0; j < PROJ_ROLE_MAXG; ++j) {
      const int i = tid + j * 1024;
      lrow[j] = i / G; lgr[j] = i - lrow[j] * G;
    }
    const bool rb = PR.bf16 != 0;
    auto zero_tail = [&](f32x4 v, int gr) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        v[c] = (4 * gr + c < K) ? v[c] : 0.0f;
        if (rb) v[c] = (float)(__bf16)v[c];
      }
      return v;
    };
    auto load_x = [&](int t, f32x4 (&rx)[PROJ_ROLE_MAXG]) {
#pragma unroll
      for (int j = 0; j < PROJ_ROLE_MAXG; ++j) {
        const bool ok = lrow[j] < B && 4 * lgr[j] < K;      // (rows >= 32 fall out here as well: B <= 32)
        const int off = ok ? (((t * B + lrow[j]) * PR.ldx + E.k_off + 4 * lgr[j]) * 4) : -16;
        rx[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, off, 0, 0));
      }
    };
    auto park = [&](float* img, const f32x4 (&rv)[PROJ_ROLE_MAXG]) {
#pragma unroll
      for (int j = 0; j < PROJ_ROLE_MAXG; ++j)
        if (lrow[j] < PROJ_ROLE_CB) *reinterpret_cast<f32x4*>(img + lrow[j] * ks + 4 * lgr[j]) = zero_tail(rv[j], lgr[j]);
    };

    f32x4 rw[PROJ_ROLE_MAXG], rx[PROJ_ROLE_MAXG];
#pragma unroll
    for (int j = 0; j < PROJ_ROLE_MAXG; ++j) {
      const int gc = col0 + lrow[j];                 // padded gate column g * Hp + u
      const int g = gc / Hp, u = gc - g * Hp;
      const bool ok = lrow[j] < PROJ_ROLE_CB && u < h && 4 * lgr[j] < K;
      const int off = ok ? (((g * h + u) * K + 4 * lgr[j]) * 4) : -16;
      rw[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, off, 0, 0));
    }
    load_x(grp, rx);
    // output element of this thread in the joining pass: (row, column) of the 32 x 32 tile
    const int orow = tid >> 5, ocol = tid & 31;
    float bias = 0.0f;
    bool cvalid;
    {
      const int gc = col0 + ocol;
      const int g = gc / Hp, u = gc - g * Hp;
      cvalid = u < h;
      if (cvalid) bias = E.b_ih[g * h + u] + E.b_hh[g * h + u];
    }
    if (grp == 0 && tid < PR.loss_n) st_agent(PR.loss_ptr + tid, 0.0f);
    park(Ws, rw);
    park(Xs, rx);
    __syncthreads();
    LSTAMP(0, 1);

    const int bi = lane & 15, q = lane >> 4;
    const int frag = wave >> 2, kpart = wave & 3;
    const int fr = frag >> 1, fc = frag & 1;
    const int NG16 = G >> 2;
    // partial (row, col) of fragment (fr, fc) lives in lane (col & 15) + 16 ((row & 15) >> 2), register row & 3
    const int pw0 = (((orow >> 4) * 2 + (ocol >> 4)) * 4) * PROJ_ROLE_PW + (orow & 3) * 96 + (ocol & 15) + 16 * ((orow & 15) >> 2);
    float* const out = d.gates;
    const int64_t orow_stride = 4 * (int64_t)Hp;
    for (int t = grp; t < T; t += PR.groups) {
      const int tn = t + PR.groups;
      if (tn < T) load_x(tn, rx);
      const float* xa = Xs + (16 * fr + bi) * ks + 4 * q;
      const float* wb = Ws + (16 * fc + bi) * ks + 4 * q;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int g16 = kpart; g16 < NG16; g16 += 4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(xa + 16 * g16);
        const f32x4 b = *reinterpret_cast<const f32x4*>(wb + 16 * g16);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc = mma16x16x4(a[c], b[c], acc);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) Ps[wave * PROJ_ROLE_PW + c * 96 + lane] = acc[c];
      __syncthreads();
      if (orow < B) {
        const float v = Ps[pw0] + Ps[pw0 + PROJ_ROLE_PW] + Ps[pw0 + 2 * PROJ_ROLE_PW] + Ps[pw0 + 3 * PROJ_ROLE_PW];
        st_agent(out + ((int64_t)t * B + orow) * orow_stride + col0 + ocol, cvalid ? v + bias : 0.0f);
      }
      if (tn < T) park(Xs, rx);                              // (every wave is past its product: the barrier above)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every store of this thread has been acknowledged
      __syncthreads();
      if (tid == 0 && !(PR.fault && r == 0 && t == grp))
        __hip_atomic_store(PR.flags + ((int64_t)e * T + t) * PROJ_ROLE_FLAGS + cbl, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      LSTAMP(0, 16 + (t - grp) / PR.groups);
    }
  }
  LSTAMP(0, 12);
  // transposed-weight images for the BPTT launches of this step (lstm_seq_dev.h)
  // (the decoders' forward images first: the decoder launch is next in the stream, the BPTT launches come later)
  wf_img_write(PR.wf, PR.n_wf, r, PR.n_role);
  wt_img_write(PR.wt, PR.n_wt, r, PR.n_role);
  LSTAMP(0, 13);
  // the launch's zero spans, spread over the role workgroups
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int s = 0; s < MFM_GEMM_ZSPANS; ++s) {
    f32x4* p = reinterpret_cast<f32x4*>(PR.zs.ptr[s]);
    const int64_t n4 = PR.zs.n[s] >> 2;
    for (int64_t i = (int64_t)r * 1024 + tid; i < n4; i += (int64_t)PR.n_role * 1024) p[i] = z4;
  }
}

}  // namespace mfm
