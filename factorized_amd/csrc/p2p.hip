// Gradient all-reduce of the data-parallel step as ONE kernel over peer-mapped staging buffers (xGMI),
// one process per GPU ("RCCL allReduce(sum) over xGMI (or the direct one-shot P2P
// variant)").  The flat gradient buffer is 1.9 MB: a ring all-reduce at that size is all latency
// (2(W-1) hops), while every MI355X has a direct link to each of its 7 peers, so a two-shot exchange
// moves 1/W of the buffer over each link twice and needs two flag rounds:
//
//   push 1   rank r writes slice s of its gradients into rank s's staging row r            (7 links, 1/W each)
//   reduce   rank s sums its W staging rows in rank order (every slice is summed exactly once, by its
//            owner, so all ranks end with bit-identical results)
//   push 2   rank s writes the reduced slice into every peer's result row s and into its own buffer
//   gather   every rank copies the W-1 foreign result rows into its gradient buffer
//
// Synchronisation is workgroup-to-workgroup, not grid-wide: workgroup w of every rank owns chunk w of every
// slice, signals `flag[r][w] = epoch` at the receiving rank after a system-scope release fence, and waits
// only for the W flags of its own chunk.  The staging block is device memory allocated uncached
// (hipDeviceMallocUncached) so that peer writes are never shadowed by a stale L2 line, and exported to the
// other processes with hipIpcGetMemHandle.  Waits are bounded by the wall clock: a rank that never arrives
// makes the others give up, set an error word and return, instead of hanging the GPU.
//
// Reuse of the rows across calls needs no double buffering: a rank can only start epoch e+1's push of chunk
// w after it has seen every owner's epoch-e result flag for chunk w, which the owner raises after reading
// its staging rows; and an owner's epoch-e+1 result push comes after every rank's epoch-e+1 first push,
// hence after that rank finished copying the epoch-e result rows.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "internal.h"

namespace mfm {

constexpr int P2P_MAXR = 8;
constexpr int P2P_WGS = 64;
constexpr int P2P_THREADS = 512;

struct P2PDev {
  float* stage[P2P_MAXR];   // per rank: [W][SL] rows written by the peers (own entry = local pointer)
  float* res[P2P_MAXR];     // per rank: [W][SL] reduced slices
  int* f1[P2P_MAXR];        // per rank: [W][P2P_WGS] epoch of the last complete first push
  int* f2[P2P_MAXR];        // per rank: [W][P2P_WGS] epoch of the last complete result push
  int* err;                 // local error word
  long long* stats;         // local: [0] ticks workgroup 0 spent in flag round 1, [1] in round 2, [2] calls (mfm_p2p_wait_stats)
  int W, rank;
  int64_t SL, CL;           // slice length, chunk length (floats, multiples of 4)
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 load4_bounded(const float* p, int64_t idx, int64_t n) {
  if (idx + 4 <= n) return *reinterpret_cast<const f32x4*>(p + idx);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < 4; ++c)
    if (idx + c < n) v[c] = p[idx + c];
  return v;
}
__device__ __forceinline__ void store4_bounded(float* p, int64_t idx, int64_t n, f32x4 v) {
  if (idx + 4 <= n) { *reinterpret_cast<f32x4*>(p + idx) = v; return; }
  for (int c = 0; c < 4; ++c)
    if (idx + c < n) p[idx + c] = v[c];
}

// Staging rows are written and read with system-scope accesses (sc0 sc1: past this GPU's L2 in both directions) on top of the
// uncached allocation.  Round 4: as ONE 16-byte instruction each -- raw buffer loads / stores carry the scope bits in their
// cache-policy operand (bit 0 = sc0, bit 4 = sc1), which HIP's scoped atomics (8 bytes at most: two instructions per float4)
// cannot express for 16 bytes.  No atomicity is needed: the flags order producer and consumer.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr int P2P_SYS = 1 | 16;            // sc0 | sc1
__device__ __forceinline__ __amdgpu_buffer_rsrc_t p2p_rsrc(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
}
// `base` must be wave-uniform (a staging block: it becomes the descriptor), `elem` is this lane's element offset inside it
__device__ __forceinline__ void store4_sys(float* base, int64_t elem, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), p2p_rsrc(base), (int)(elem * 4), 0, P2P_SYS);
}
__device__ __forceinline__ f32x4 load4_sys(const float* base, int64_t elem) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(p2p_rsrc(base), (int)(elem * 4), 0, P2P_SYS));
}

// A flag word is 2 * epoch + bad: the low bit of a FIRST-push flag says that the sending rank's gradients carry a raised guard
// word (mfm_p2p_allreduce_adam_guarded: a hand-over of its step gave up, include/mfm_hip.h) -- every workgroup of every rank
// sees the bits of all ranks before it touches a parameter, so all replicas skip the same steps.
// thread t < W waits until flags[t] reaches `epoch`; returns (after a block barrier + acquire fence) whether any flag had its
// low bit set
__device__ __forceinline__ bool wait_flags(const int* flags, int stride, int W, int epoch, int64_t timeout, int* err,
                                           long long* stat = nullptr) {
  __shared__ int any_bad;
  __shared__ int waited;
  if (threadIdx.x == 0) { any_bad = 0; waited = 0; }
  __syncthreads();
  if ((int)threadIdx.x < W) {
    const int* f = flags + (int64_t)threadIdx.x * stride;
    const int64_t t0 = wall_clock64();
    int v;
    while (((v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) >> 1) < epoch) {
      __builtin_amdgcn_s_sleep(2);
      // give up after `timeout`, and at once when an earlier wait of this rank already did
      if (wall_clock64() - t0 > timeout || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        atomicExch(err, 1);
        v = 0;
        break;
      }
    }
    if (v & 1) any_bad = 1;
    if (stat) atomicMax(&waited, (int)min((int64_t)0x7fffffff, wall_clock64() - t0));
  }
  // one acquire per workgroup (the wave that polled): it invalidates this CU's vector L1 and the L2's
  // non-coherent lines; the staging block itself is uncached, so this is insurance, not the mechanism
#ifndef P2P_EXPERIMENT_NO_SYSTEM_FENCE
  if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
#endif
  __syncthreads();
  // diagnosis (bench.py --gpus N): how long workgroup 0 of this rank spun in this flag round (100 MHz ticks, summed over calls;
  // one writer, read by the host between launches)
  if (stat && threadIdx.x == 0) *stat += waited;
  return any_bad != 0;
}

// every wave waits for its own stores to be acknowledged, then ONE wave does the system-scope release (an L2
// write-back on gfx950: once per workgroup, not once per wave) and raises the flags
__device__ __forceinline__ void publish(int* const* flags, int slot, int W, int epoch, int bad = 0) {
  // (explicit: a workgroup-scope release needs no vmcnt wait on gfx950 outside threadgroup-split mode, so the fence alone
  // would let the flag below overtake the other waves' stores -- the hole found in the role-workgroup hand-overs, round 4)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x < 64) {
#ifndef P2P_EXPERIMENT_NO_SYSTEM_FENCE
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
#endif
    if ((int)threadIdx.x < W) __hip_atomic_store(flags[threadIdx.x] + slot, 2 * epoch + bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Optional fused optimizer: the rank that holds a reduced gradient in registers applies the flat Adam update
// (same arithmetic as adam_kernel, elementwise.hip) instead of a separate launch reading it back.
struct AdamArgs {
  float *p, *m, *v;
  float beta1, beta2, eps, step_size, bc2_sqrt, grad_scale;
};

__device__ __forceinline__ void adam4(const AdamArgs& a, int64_t idx, int64_t n, f32x4 g4) {
  f32x4 pv = load4_bounded(a.p, idx, n), mv = load4_bounded(a.m, idx, n), vv = load4_bounded(a.v, idx, n);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float gg = g4[j] * a.grad_scale;
    mv[j] = mv[j] + (1.0f - a.beta1) * (gg - mv[j]);
    vv[j] = vv[j] * a.beta2 + (1.0f - a.beta2) * gg * gg;
    const float denom = sqrtf(vv[j]) / a.bc2_sqrt + a.eps;
    pv[j] = pv[j] - a.step_size * mv[j] / denom;
  }
  store4_bounded(a.p, idx, n, pv);
  store4_bounded(a.m, idx, n, mv);
  store4_bounded(a.v, idx, n, vv);
}

template <bool ADAM>
__global__ __launch_bounds__(P2P_THREADS) void p2p_allreduce_kernel(const P2PDev d, float* __restrict__ buf, int64_t n, int epoch,
                                                                   int64_t timeout, const AdamArgs ad, const int64_t guard_idx) {
  const int w = blockIdx.x, tid = threadIdx.x;
  const int W = d.W, me = d.rank;
  // this rank's guard word (written by an earlier launch: the same value in every workgroup)
  const int bad_local = (ADAM && guard_idx >= 0 && !(buf[guard_idx] == 0.0f)) ? 1 : 0;
  const int64_t SL = d.SL, CL = d.CL, cb = (int64_t)w * CL;
  // All loops over ranks are unrolled to P2P_MAXR with clamped indices and predicated stores: the loads of
  // one pass are then issued back to back instead of one dependent round trip per rank.
  // ---- push 1: my chunk w of every slice -> the slice owner's staging row `me`
  for (int64_t k = (int64_t)tid * 4; k < CL && cb + k < SL; k += P2P_THREADS * 4) {
    f32x4 v[P2P_MAXR];
#pragma unroll
    for (int i = 0; i < P2P_MAXR; ++i) {
      const int s = (me + 1 + min(i, W - 1)) % W;         // start at the next rank: spreads the links
      v[i] = load4_bounded(buf, (int64_t)s * SL + cb + k, n);
    }
#pragma unroll
    for (int i = 0; i < P2P_MAXR; ++i) {
      const int s = (me + 1 + min(i, W - 1)) % W;
      if (i < W) store4_sys(d.stage[s], (int64_t)me * SL + cb + k, v[i]);
    }
  }
  publish(d.f1, me * P2P_WGS + w, W, epoch, bad_local);
  // ---- reduce my slice's chunk in rank order, push 2: the result -> every rank's result row `me`
  long long* stat = (w == 0) ? d.stats : nullptr;
  const bool skip = wait_flags(d.f1[me] + w, P2P_WGS, W, epoch, timeout, d.err, stat);
  // The result goes to the peers FIRST and the flags right behind it; this rank's own copy and its Adam update -- plain stores,
  // i.e. dirty L2 lines that the release inside publish() would have to write back before the flags could leave -- come after
  // (round 4; while the chunk fits two passes of the workgroup: 4096 floats, every gradient buffer of this code base at W >= 2)
  const bool early = CL <= 2 * P2P_THREADS * 4;
  f32x4 keep[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  {
    int pass = 0;
    for (int64_t k = (int64_t)tid * 4; k < CL && cb + k < SL; k += P2P_THREADS * 4, ++pass) {
      f32x4 v[P2P_MAXR];
#pragma unroll
      for (int r = 0; r < P2P_MAXR; ++r) v[r] = load4_sys(d.stage[me], cb + k + (int64_t)min(r, W - 1) * SL);
      f32x4 acc = v[0];
#pragma unroll
      for (int r = 1; r < P2P_MAXR; ++r)
        if (r < W) acc += v[r];
#pragma unroll
      for (int i = 1; i < P2P_MAXR; ++i) {
        const int p = (me + min(i, W - 1)) % W;
        if (i < W) store4_sys(d.res[p], (int64_t)me * SL + cb + k, acc);
      }
      if (early) { if (pass == 0) keep[0] = acc; else keep[1] = acc; }
      else {
        store4_bounded(buf, (int64_t)me * SL + cb + k, n, acc);
        if (ADAM && !skip) adam4(ad, (int64_t)me * SL + cb + k, n, acc);
      }
    }
  }
  publish(d.f2, me * P2P_WGS + w, W, epoch);
  if (early) {
    int pass = 0;
    for (int64_t k = (int64_t)tid * 4; k < CL && cb + k < SL; k += P2P_THREADS * 4, ++pass) {
      const f32x4 acc = pass == 0 ? keep[0] : keep[1];
      store4_bounded(buf, (int64_t)me * SL + cb + k, n, acc);
      if (ADAM && !skip) adam4(ad, (int64_t)me * SL + cb + k, n, acc);
    }
  }
  // ---- gather: foreign result rows -> my gradient buffer
  (void)wait_flags(d.f2[me] + w, P2P_WGS, W, epoch, timeout, d.err, stat ? stat + 1 : nullptr);
  if (stat && tid == 0) stat[2] += 1;
  for (int64_t k = (int64_t)tid * 4; k < CL && cb + k < SL; k += P2P_THREADS * 4) {
    f32x4 v[P2P_MAXR];
#pragma unroll
    for (int i = 1; i < P2P_MAXR; ++i) {
      const int s = (me + min(i, W - 1)) % W;
      v[i] = load4_sys(d.res[me], (int64_t)s * SL + cb + k);
    }
#pragma unroll
    for (int i = 1; i < P2P_MAXR; ++i) {
      const int s = (me + min(i, W - 1)) % W;
      if (i < W) {
        store4_bounded(buf, (int64_t)s * SL + cb + k, n, v[i]);
        if (ADAM && !skip) adam4(ad, (int64_t)s * SL + cb + k, n, v[i]);
      }
    }
  }
}

// The same exchange when a chunk is at most TWO passes of the workgroup and ranks x passes <= 8 (the 1.9 MB gradient buffer at
// every W from 2 to 8: one pass from W = 4 on, two at W = 2) -- straight-line code whose memory round trips are taken TOGETHER
// instead of one behind the other (round 5):
//   * the Adam operands p, m, v of everything this workgroup will update (its chunk of all W slices) are requested right behind
//     the first push, so they travel while the flags do; the generic kernel asked for them slice by slice behind the second flag
//     round, each request waiting for the previous slice's stores (no-alias cannot be proven): W - 1 dependent round trips;
//   * the W - 1 foreign result rows are requested in one batch and updated from registers.
// NP = passes, MAXW = ranks the register arrays are sized for (NP * MAXW = 8 items of p, m, v: 96 registers).
template <bool ADAM, int NP, int MAXW>
__global__ __launch_bounds__(P2P_THREADS) void p2p_allreduce_flat_kernel(const P2PDev d, float* __restrict__ buf, int64_t n, int epoch,
                                                                        int64_t timeout, const AdamArgs ad, const int64_t guard_idx) {
  constexpr int NI = NP * MAXW;
  const int w = blockIdx.x, tid = threadIdx.x;
  const int W = d.W, me = d.rank;
  const int bad_local = (ADAM && guard_idx >= 0 && !(buf[guard_idx] == 0.0f)) ? 1 : 0;
  const int64_t SL = d.SL, CL = d.CL, cb = (int64_t)w * CL;
  long long* stat = (w == 0) ? d.stats : nullptr;
  // item (i, q): slice visited at position i (the own slice at i = 0, then the peers starting at the next rank: spreads the
  // links), pass q of the chunk
  auto slice = [&](int i) { return (me + min(i, W - 1)) % W; };
  auto koff = [&](int q) { return (int64_t)tid * 4 + (int64_t)q * P2P_THREADS * 4; };
  auto act = [&](int q) { return koff(q) < CL && cb + koff(q) < SL; };
  f32x4 v[NI];
#pragma unroll
  for (int i = 0; i < MAXW; ++i)
#pragma unroll
    for (int q = 0; q < NP; ++q)
      if (act(q)) v[i * NP + q] = load4_bounded(buf, (int64_t)slice(i) * SL + cb + koff(q), n);
#pragma unroll
  for (int i = 0; i < MAXW; ++i)
#pragma unroll
    for (int q = 0; q < NP; ++q)
      if (i < W && act(q)) store4_sys(d.stage[slice(i)], (int64_t)me * SL + cb + koff(q), v[i * NP + q]);
  f32x4 pp[NI], pm[NI], pv[NI];
  if (ADAM) {
#pragma unroll
    for (int i = 0; i < MAXW; ++i)
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        if (!act(q)) continue;
        const int64_t idx = (int64_t)slice(i) * SL + cb + koff(q);
        pp[i * NP + q] = load4_bounded(ad.p, idx, n); pm[i * NP + q] = load4_bounded(ad.m, idx, n); pv[i * NP + q] = load4_bounded(ad.v, idx, n);
      }
  }
  publish(d.f1, me * P2P_WGS + w, W, epoch, bad_local);
  const bool skip = wait_flags(d.f1[me] + w, P2P_WGS, W, epoch, timeout, d.err, stat);
  auto adam_from = [&](int it, int64_t idx, f32x4 g4) {
    f32x4 p4 = pp[it], m4 = pm[it], v4 = pv[it];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gg = g4[j] * ad.grad_scale;
      m4[j] = m4[j] + (1.0f - ad.beta1) * (gg - m4[j]);
      v4[j] = v4[j] * ad.beta2 + (1.0f - ad.beta2) * gg * gg;
      const float denom = sqrtf(v4[j]) / ad.bc2_sqrt + ad.eps;
      p4[j] = p4[j] - ad.step_size * m4[j] / denom;
    }
    store4_bounded(ad.p, idx, n, p4); store4_bounded(ad.m, idx, n, m4); store4_bounded(ad.v, idx, n, v4);
  };
  // ---- reduce my slice's chunk in rank order, push 2: the result -> every rank's result row `me`
  f32x4 acc[NP];
  {
    f32x4 r[NI];
#pragma unroll
    for (int rk = 0; rk < MAXW; ++rk)
#pragma unroll
      for (int q = 0; q < NP; ++q)
        if (act(q)) r[rk * NP + q] = load4_sys(d.stage[me], cb + koff(q) + (int64_t)min(rk, W - 1) * SL);
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (!act(q)) continue;
      acc[q] = r[q];
#pragma unroll
      for (int rk = 1; rk < MAXW; ++rk)
        if (rk < W) acc[q] += r[rk * NP + q];
#pragma unroll
      for (int i = 1; i < MAXW; ++i)
        if (i < W) store4_sys(d.res[slice(i)], (int64_t)me * SL + cb + koff(q), acc[q]);
    }
  }
  publish(d.f2, me * P2P_WGS + w, W, epoch);
  // (own copy and own update behind the flags: plain stores = dirty L2 lines the release above would have had to write back)
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    if (!act(q)) continue;
    const int64_t idx = (int64_t)me * SL + cb + koff(q);
    store4_bounded(buf, idx, n, acc[q]);
    if (ADAM && !skip) adam_from(q, idx, acc[q]);
  }
  (void)wait_flags(d.f2[me] + w, P2P_WGS, W, epoch, timeout, d.err, stat ? stat + 1 : nullptr);
  // ---- gather: foreign result rows -> my gradient buffer (+ Adam from the prefetched operands)
#pragma unroll
  for (int i = 1; i < MAXW; ++i)
#pragma unroll
    for (int q = 0; q < NP; ++q)
      if (act(q)) v[i * NP + q] = load4_sys(d.res[me], (int64_t)slice(i) * SL + cb + koff(q));
#pragma unroll
  for (int i = 1; i < MAXW; ++i)
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      if (i >= W || !act(q)) continue;
      const int64_t idx = (int64_t)slice(i) * SL + cb + koff(q);
      store4_bounded(buf, idx, n, v[i * NP + q]);
      if (ADAM && !skip) adam_from(i * NP + q, idx, v[i * NP + q]);
    }
  if (stat && tid == 0) stat[2] += 1;
}

__global__ __launch_bounds__(P2P_THREADS) void adam_only_kernel(const float* __restrict__ g, int64_t n, const AdamArgs ad,
                                                                const int64_t guard_idx) {
  if (guard_idx >= 0 && !(g[guard_idx] == 0.0f)) return;
  for (int64_t k = ((int64_t)blockIdx.x * P2P_THREADS + threadIdx.x) * 4; k < n; k += (int64_t)P2P_WGS * P2P_THREADS * 4)
    adam4(ad, k, n, load4_bounded(g, k, n));
}

struct P2P {
  P2PDev d;
  void* local = nullptr;
  void* peer[P2P_MAXR] = {};
  size_t bytes = 0, off_res = 0, off_f1 = 0, off_f2 = 0, off_err = 0;
  int64_t max_elems = 0;
  int epoch = 0;
  int64_t timeout_ticks = 0;
  bool connected = false;
};

static void p2p_point(P2P* h, int r, void* base) {
  char* b = static_cast<char*>(base);
  h->d.stage[r] = reinterpret_cast<float*>(b);
  h->d.res[r] = reinterpret_cast<float*>(b + h->off_res);
  h->d.f1[r] = reinterpret_cast<int*>(b + h->off_f1);
  h->d.f2[r] = reinterpret_cast<int*>(b + h->off_f2);
}

}  // namespace mfm

using namespace mfm;

extern "C" {

int mfm_p2p_create(int32_t nranks, int32_t rank, int64_t max_elems, void** handle) {
  if (!handle || nranks < 1 || nranks > P2P_MAXR || rank < 0 || rank >= nranks || max_elems < 1) {
    set_error("mfm_p2p_create: need 1 <= nranks <= %d, 0 <= rank < nranks, max_elems >= 1", P2P_MAXR);
    return MFM_ERR_ARG;
  }
  P2P* h = new P2P();
  h->max_elems = max_elems;
  h->d.W = nranks;
  h->d.rank = rank;
  const int64_t per = (max_elems + nranks - 1) / nranks;
  h->d.CL = ((per + P2P_WGS - 1) / P2P_WGS + 3) / 4 * 4;
  h->d.SL = ((per + 3) / 4) * 4;
  auto up = [](size_t v) { return (v + 4095) / 4096 * 4096; };
  const size_t rows = up((size_t)nranks * h->d.SL * sizeof(float));
  const size_t flags = up((size_t)nranks * P2P_WGS * sizeof(int));
  h->off_res = rows;
  h->off_f1 = 2 * rows;
  h->off_f2 = 2 * rows + flags;
  h->off_err = 2 * rows + 2 * flags;
  h->bytes = h->off_err + 4096;
  hipError_t e = hipExtMallocWithFlags(&h->local, h->bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipExtMallocWithFlags(&h->local, h->bytes, hipDeviceMallocFinegrained);
  }
  if (e != hipSuccess) { delete h; return hip_fail(e, "mfm_p2p_create: staging allocation"); }
  e = hipMemset(h->local, 0, h->bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) { (void)hipFree(h->local); delete h; return hip_fail(e, "mfm_p2p_create: memset"); }
  p2p_point(h, rank, h->local);
  h->peer[rank] = h->local;
  h->d.err = reinterpret_cast<int*>(static_cast<char*>(h->local) + h->off_err);
  h->d.stats = reinterpret_cast<long long*>(static_cast<char*>(h->local) + h->off_err + 256);
  const char* t = opt_get("MFM_P2P_TIMEOUT_MS");
  const double ms = t ? atof(t) : 10000.0;
  h->timeout_ticks = (int64_t)(ms * 1e5);               // wall_clock64 ticks at 100 MHz
  h->connected = (nranks == 1);
  *handle = h;
  return MFM_OK;
}

int mfm_p2p_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

int mfm_p2p_export(void* handle, void* out) {
  P2P* h = static_cast<P2P*>(handle);
  if (!h || !out) { set_error("mfm_p2p_export: null argument"); return MFM_ERR_ARG; }
  hipIpcMemHandle_t ipc;
  hipError_t e = hipIpcGetMemHandle(&ipc, h->local);
  if (e != hipSuccess) return hip_fail(e, "hipIpcGetMemHandle");
  memcpy(out, &ipc, sizeof(ipc));
  return MFM_OK;
}

int mfm_p2p_connect(void* handle, const void* all_handles) {
  P2P* h = static_cast<P2P*>(handle);
  if (!h || !all_handles) { set_error("mfm_p2p_connect: null argument"); return MFM_ERR_ARG; }
  // (peer access to a block that lives on another GPU is enabled by the open call itself)
  for (int r = 0; r < h->d.W; ++r) {
    if (r == h->d.rank) continue;
    hipIpcMemHandle_t ipc;
    memcpy(&ipc, static_cast<const char*>(all_handles) + (size_t)r * sizeof(ipc), sizeof(ipc));
    void* base = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&base, ipc, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return hip_fail(e, "hipIpcOpenMemHandle");
    h->peer[r] = base;
    p2p_point(h, r, base);
  }
  h->connected = true;
  return MFM_OK;
}

void* mfm_p2p_local_base(void* handle) {
  P2P* h = static_cast<P2P*>(handle);
  return h ? h->local : nullptr;
}

int mfm_p2p_connect_bases(void* handle, const void* const* bases) {
  P2P* h = static_cast<P2P*>(handle);
  if (!h || !bases) { set_error("mfm_p2p_connect_bases: null argument"); return MFM_ERR_ARG; }
  for (int r = 0; r < h->d.W; ++r) {
    if (r == h->d.rank) continue;
    if (!bases[r]) { set_error("mfm_p2p_connect_bases: base of rank %d is null", r); return MFM_ERR_ARG; }
    p2p_point(h, r, const_cast<void*>(bases[r]));        // not owned: peer[] stays empty, nothing to close
  }
  h->connected = true;
  return MFM_OK;
}

static int p2p_launch(void* handle, float* buf, int64_t n, void* stream, const AdamArgs* ad, const char* who, int64_t guard_idx = -1) {
  P2P* h = static_cast<P2P*>(handle);
  if (!h || !buf) { set_error("%s: null argument", who); return MFM_ERR_ARG; }
  if (!h->connected) { set_error("%s: mfm_p2p_connect has not been called", who); return MFM_ERR_ARG; }
  if (n < 0 || n > h->max_elems) { set_error("%s: n=%lld exceeds max_elems=%lld", who, (long long)n, (long long)h->max_elems); return MFM_ERR_ARG; }
  if ((reinterpret_cast<uintptr_t>(buf) & 15) != 0) { set_error("%s: buffer must be 16-byte aligned", who); return MFM_ERR_ARG; }
  if (n == 0) return MFM_OK;
  if (guard_idx >= n) { set_error("%s: guard index %lld outside the buffer of %lld elements", who, (long long)guard_idx, (long long)n); return MFM_ERR_ARG; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (h->d.W == 1) {            // one rank: the sum is the buffer itself; only the optimizer is left to do
    if (!ad) return MFM_OK;
    hipLaunchKernelGGL(adam_only_kernel, dim3(P2P_WGS), dim3(P2P_THREADS), 0, st, buf, n, *ad, guard_idx);
  } else {
    h->epoch += 1;
    // chunks of one / two workgroup passes with ranks x passes <= 8: the flat kernel (every round trip batched)
    const int np = h->d.CL <= (int64_t)P2P_THREADS * 4 ? 1 : (h->d.CL <= (int64_t)2 * P2P_THREADS * 4 ? 2 : 0);
    const int flat = opt_get("MFM_P2P_GENERIC") ? 0 : (np == 1 ? 1 : (np == 2 && h->d.W <= 4 ? 2 : 0));
#define P2P_FLAT_LAUNCH(A, NPV, MW, ADV, GI) \
    hipLaunchKernelGGL((p2p_allreduce_flat_kernel<A, NPV, MW>), dim3(P2P_WGS), dim3(P2P_THREADS), 0, st, h->d, buf, n, h->epoch, h->timeout_ticks, ADV, GI)
    if (flat == 1 && ad) P2P_FLAT_LAUNCH(true, 1, 8, *ad, guard_idx);
    else if (flat == 1) P2P_FLAT_LAUNCH(false, 1, 8, AdamArgs{}, (int64_t)-1);
    else if (flat == 2 && ad) P2P_FLAT_LAUNCH(true, 2, 4, *ad, guard_idx);
    else if (flat == 2) P2P_FLAT_LAUNCH(false, 2, 4, AdamArgs{}, (int64_t)-1);
#undef P2P_FLAT_LAUNCH
    else if (ad)
      hipLaunchKernelGGL(p2p_allreduce_kernel<true>, dim3(P2P_WGS), dim3(P2P_THREADS), 0, st, h->d, buf, n, h->epoch, h->timeout_ticks, *ad, guard_idx);
    else
      hipLaunchKernelGGL(p2p_allreduce_kernel<false>, dim3(P2P_WGS), dim3(P2P_THREADS), 0, st, h->d, buf, n, h->epoch, h->timeout_ticks,
                         AdamArgs{}, (int64_t)-1);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, who);
  return MFM_OK;
}

int mfm_p2p_allreduce(void* handle, float* buf, int64_t n, void* stream) {
  return p2p_launch(handle, buf, n, stream, nullptr, "mfm_p2p_allreduce");
}

int mfm_p2p_allreduce_adam(void* handle, float* grads, float* p, float* m, float* v, int64_t n, int32_t step, float lr, float beta1,
                           float beta2, float eps, float grad_scale, void* stream) {
  return mfm_p2p_allreduce_adam_guarded(handle, grads, p, m, v, n, step, lr, beta1, beta2, eps, grad_scale, -1, stream);
}

int mfm_p2p_allreduce_adam_guarded(void* handle, float* grads, float* p, float* m, float* v, int64_t n, int32_t step, float lr,
                                   float beta1, float beta2, float eps, float grad_scale, int64_t guard_index, void* stream) {
  if (!p || !m || !v || step < 1) { set_error("mfm_p2p_allreduce_adam: bad arguments (step=%d)", step); return MFM_ERR_ARG; }
  if (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) != 0) {
    set_error("mfm_p2p_allreduce_adam: buffers must be 16-byte aligned");
    return MFM_ERR_ARG;
  }
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  AdamArgs ad{p, m, v, beta1, beta2, eps, (float)((double)lr / bc1), (float)sqrt(bc2), grad_scale};
  return p2p_launch(handle, grads, n, stream, &ad, "mfm_p2p_allreduce_adam", guard_index);
}

int mfm_p2p_status(void* handle, int32_t* timed_out) {
  P2P* h = static_cast<P2P*>(handle);
  if (!h || !timed_out) { set_error("mfm_p2p_status: null argument"); return MFM_ERR_ARG; }
  int v = 0;
  hipError_t e = hipMemcpy(&v, h->d.err, sizeof(int), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return hip_fail(e, "mfm_p2p_status: read-back");
  *timed_out = v;
  return MFM_OK;
}

int mfm_p2p_wait_stats(void* handle, int64_t* out, int32_t reset) {
  P2P* h = static_cast<P2P*>(handle);
  if (!h || !out) { set_error("mfm_p2p_wait_stats: null argument"); return MFM_ERR_ARG; }
  long long v[3] = {0, 0, 0};
  hipError_t e = hipMemcpy(v, h->d.stats, sizeof(v), hipMemcpyDeviceToHost);
  if (e == hipSuccess && reset) e = hipMemset(h->d.stats, 0, sizeof(v));
  if (e != hipSuccess) return hip_fail(e, "mfm_p2p_wait_stats");
  out[0] = v[0]; out[1] = v[1]; out[2] = v[2];
  return MFM_OK;
}

void mfm_p2p_destroy(void* handle) {
  P2P* h = static_cast<P2P*>(handle);
  if (!h) return;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < h->d.W; ++r)
    if (r != h->d.rank && h->peer[r]) (void)hipIpcCloseMemHandle(h->peer[r]);
  if (h->local) (void)hipFree(h->local);
  delete h;
}

}  // extern "C"
