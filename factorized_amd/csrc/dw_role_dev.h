// Weight gradients as ROLE workgroups of the encoder BPTT launch (round 3, MFM_KL_EF at B <= 32).
//
// Every weight-gradient product of the step is a sum over rows, C[m, n] += sum_r A[r, m] B[r, n] (gemm_tn.hip), and at the
// reference's batch size they were a launch of their own behind the encoder BPTT: ~18 us plus the boundary, while the BPTT
// itself keeps only 4 B <= 128 of the 256 CUs busy for ~43 us.  Here the idle CUs run those products INSIDE the BPTT launch:
//   * a role workgroup (1024 threads) is four SLOTS of 256 threads, each with the arithmetic of gemm_tn_kernel on one
//     (tile, 128-row chunk) block per iteration: whole operand slices requested at once, 32 MFMA steps per wave; the four
//     slots of a workgroup hold tiles that share their A slice (round 4), which is fetched once;
//   * a device table built once per plan gives every slot its block per iteration ([iteration][slot]), ordered by when the
//     A operand becomes final: first the decoder-side products (final before the launch) and the latent stack's (after the
//     rows' chains at the head of this launch), one chunk per block, partial tile added with atomics; then every ENCODER
//     tile stays with one slot for all its chunks, from the last time steps to the first, accumulates in registers and
//     is added to the gradient once (the launch it replaces spent its time on ~1.5 M memory-side atomics);
//   * the BPTT workgroup of (encoder e, row b) writes dA_t with agent-scope stores and, one step later -- when those stores
//     are acknowledged anyway (s_waitcnt vmcnt(#loads of the newer prefetch)) -- stamps flags[e][t][b] with the launch's
//     epoch; a block whose chunk starts at time step t0 polls the B stamps of (e, t0) and reads dA with agent-scope loads.
// No deadlock: the BPTT workgroups have the lower block ids and never wait.  What stays exposed is the last chunk
// (time steps 0..3): one round of blocks behind the end of the BPTT.
#pragma once
#include "internal.h"
#include "lstm_seq_dev.h"
#include "proj_role_dev.h"
#include "lstamp.h"

namespace mfm {

constexpr int DWR_KC = 128;            // rows per chunk
constexpr int DWR_T = 32;              // tile edge
constexpr int DWR_MAXP = 56;
constexpr int DWR_ROWS = 64;           // stamp words per (encoder, time step): B <= 64 (round 4; the projections' role form stays at B <= 32)
constexpr int DWR_TABLE_CAP = 16384;   // table entries carved in the plan workspace

struct DwRoleProblem {
  const float* a; const float* b; float* c; float* c2;
  int a_sz, b_sz, c_sz, a_sk, b_sk, ldc, m, n_valid, k, tiles_m, tiles_n, kps, batch;
  int b_shift;                         // chunk row r pairs with row r - b_shift of B (rows r < b_shift contribute nothing)
  float alpha;
};
struct DwRole {
  DwRoleProblem p[DWR_MAXP];
  int count, n_iter, n_role, T, B, any_dep;
  int bf16;                            // bf16 plans: both operands rounded to bf16 (RNE) on the way into LDS, fp32 accumulation
  const int4* table;                   // [n_iter][4 n_role]: x = problem (-1: idle), y = tile (tn + tiles_n (tm + tiles_m z)), z = chunk,
                                       // w = dep | t0 << 8 | FIRST << 24 | LAST << 25 | accumulator << 26 | STORE << 27; the active slots
                                       // of a workgroup are a prefix and share (A operand, z, tm, chunk, dep, t0) with its slot 0
  unsigned* flags;                     // [4][T][32] BPTT stamps, then [4][B] latent-chain stamps
  unsigned epoch;                      // host part of the launch's epoch (proj_role_dev.h, ho_epoch)
  const unsigned* tick;                // device part: the plan's backward replay counter
  HoCtl ctl;                           // time-out, status word, poison (= the gradient guard: the optimizer skips the step)
  int fault;                           // fault injection (tests): the BPTT workgroup of (encoder 0, row 0) does not stamp t = 0
  int lat_split;                       // the rows' latent chains hand d h_T to their BPTT through LDS and send their stores out
                                       // while the BPTT's weights travel (lstm_seq_small_folddw_kernel)
};
// dep codes of a table entry
constexpr int DWR_DEP_NONE = 0, DWR_DEP_LATENT = 5;     // 1..4: encoder e = dep - 1
constexpr int DWR_FIRST = 1 << 24, DWR_LAST = 1 << 25, DWR_ACC1 = 1 << 26, DWR_STORE = 1 << 27;

int seq_small_folddw_launch(SeqLaunch& L, const LatentDev& LD, DwRole& DR, const float* params, float* grads, hipStream_t stream);
int seq_folddw_launch(const MfmSeqDesc* descs, int count, int T, int B, const LatentDev& lat, const float* params, float* grads,
                      DwRole& dr, hipStream_t stream, const float* const* wt_imgs = nullptr);
bool seq_small_folddw_supported(int T, int B);

__device__ __forceinline__ void dwr_stamp(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// wave-level wait for `n` stamps at f[0..n) (n <= 256)
__device__ __forceinline__ void dwr_wait(const unsigned* f, int n, unsigned epoch, const HoCtl& ctl) {
  const int lane = threadIdx.x & 63;
  const long long t0 = wall_clock64();
  for (;;) {
    const unsigned v0 = __hip_atomic_load(f + (lane < n ? lane : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned v1 = __hip_atomic_load(f + (lane + 64 < n ? lane + 64 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned v2 = epoch, v3 = epoch;
    if (n > 128) {      // (uniform)
      v2 = __hip_atomic_load(f + (lane + 128 < n ? lane + 128 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v3 = __hip_atomic_load(f + (lane + 192 < n ? lane + 192 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (__builtin_amdgcn_ballot_w64(v0 != epoch || v1 != epoch || v2 != epoch || v3 != epoch) == 0ull) return;
    if (wall_clock64() - t0 > ctl.timeout) { ho_give_up(ctl); return; }      // (default ~50 ms: proj_role_dev.h, HoCtl)
    __builtin_amdgcn_s_sleep(8);
  }
}

// non-blocking form: true when all `n` stamps carry `epoch` (wave-uniform; every lane takes part)
__device__ __forceinline__ bool dwr_ready(const unsigned* f, int n, unsigned epoch) {
  const int lane = threadIdx.x & 63;
  const unsigned v0 = __hip_atomic_load(f + (lane < n ? lane : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned v1 = __hip_atomic_load(f + (lane + 64 < n ? lane + 64 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned v2 = epoch, v3 = epoch;
  if (n > 128) {
    v2 = __hip_atomic_load(f + (lane + 128 < n ? lane + 128 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v3 = __hip_atomic_load(f + (lane + 192 < n ? lane + 192 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return __builtin_amdgcn_ballot_w64(v0 != epoch || v1 != epoch || v2 != epoch || v3 != epoch) == 0ull;
}

// One (iteration, slot) entry of the block table, resolved against the problem descriptors ONCE per launch (round 6).  The launch
// clock (profiles/r06_launch_timeline.txt) put 2.6 of the 6.6 us a table iteration takes into bookkeeping: two table loads from
// memory, then a chain of dependent scalar loads of the by-value problem descriptors for the operand addresses -- per iteration,
// in front of the loads they describe.  Now the workgroup copies the descriptors from the kernel-argument segment into LDS,
// thread e builds the record of entry e = 4 * iteration + slot, and an iteration starts with six LDS reads.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
struct DwRec {
  unsigned long long a, b, c, c2;                           // A slice (workgroup), B and C / C2 of this slot, gate block included
  int a_bytes, a_sk, m, m0, kbeg, klen, dept0, w;           // dept0 = dep | t0 << 8; w = table flags | DWR_ACTIVE | DWR_GACTIVE
  int b_bytes, b_sk, b_row0, n0, n_valid, ldc; float alpha; int pm;     // b_row0 = kbeg - b_shift; pm = rows of C
};
static_assert(sizeof(DwRec) == 96 && sizeof(DwRoleProblem) == 96, "dw role records: 24 dwords each");
static_assert(alignof(SeqLaunch) == 8 && alignof(LatentDev) == 8 && alignof(DwRole) == 8 && sizeof(SeqLaunch) % 8 == 0 && sizeof(LatentDev) % 8 == 0,
              "dw role records: the kernel-argument offset of the DwRole descriptor (lstm_seq_small_folddw_kernel)");
constexpr int DWR_MAXREC = 128;                             // entries per workgroup: DWR_TABLE_CAP / (4 * 32 role workgroups at least)
constexpr int DWR_ACTIVE = 1 << 28, DWR_GACTIVE = 1 << 29;
constexpr int DWR_LDS_FLOATS = 5 * DWR_KC * DWR_T + 16 + DWR_MAXREC * 24 + DWR_MAXP * 24;    // images, answer word, records, descriptors

// `karg_dr`: the DwRole descriptor in the kernel-argument segment (read with per-lane indices: never through the by-value copy)
__device__ __forceinline__ void dw_role_body(const DwRole& DR, const unsigned epoch, float* lds, const int* __restrict__ karg_dr) {
  // One role workgroup = four slots of 256 threads that work on tiles with the SAME A slice (same rows, same 32 columns of
  // the gate-gradient / upstream-gradient operand: the dW_ih, dW_hh and bias tiles of one gate block, the column tiles of one
  // decoder fc1 row block ...): the slice is fetched once per workgroup -- one 16-byte agent-scope load per thread -- instead
  // of once per tile (44.4 -> 39.0 MB of HBM-side reads per launch, profiles/r04_traffic_B32.json).
  // Software pipeline (round 4): a block is a chain of memory round trips (stamp poll, A from the memory side, B from L2),
  // ~5.5 us when run back to back, and a workgroup has 8-12 of them behind a 41 us BPTT.  So the NEXT block's operands are
  // requested while the current one is multiplied: B always (it never depends on this launch), A when its stamps are already
  // there (checked without blocking: the stamp loads are issued at the top of the iteration, read before its first barrier).
  // (Round 6 tried requesting A speculatively and looking at the stamps behind the product: no gain -- 48.6 us either way -- and
  // unsound as written: a flag load and a data load of one wave may be SERVICED out of order, so "the stamps I asked for earlier
  // carried the epoch" does not make a data load issued before they returned a load of final data.  The stamps are looked at
  // before A is requested.)
  constexpr int BL = DWR_KC * (DWR_T / 4) / 256;       // 16-byte loads per thread for a slot's B slice
  const int tid = threadIdx.x, sub = tid >> 8, t = tid & 255;
  const int lane = t & 63, wave = t >> 6;
  const int bi = lane & 15, q = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  float* As = lds;                                                 // [DWR_KC][32], shared
  float* Bs = lds + (1 + sub) * (DWR_KC * DWR_T);                  // [DWR_KC][32] per slot
  int* nready_lds = reinterpret_cast<int*>(lds + 5 * DWR_KC * DWR_T);
  int* recs = reinterpret_cast<int*>(lds + 5 * DWR_KC * DWR_T + 16);              // [n_iter][4] DwRec
  int* probs = recs + DWR_MAXREC * 24;                                             // [count] DwRoleProblem (build only)
  const int r = blockIdx.x - 4 * DR.B;           // role index
  const int nslots = 4 * DR.n_role;
  const int n_iter = DR.n_iter;
  f32x4 accs[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};

  // ---- records
  {
    const int nw = DR.count * 24;
    for (int i = tid; i < nw; i += 1024) probs[i] = karg_dr[i];              // (DwRole::p is the descriptor's first member)
    int4 ent = {-1, 0, 0, 0}, ea = {-1, 0, 0, 0};
    const int it = tid >> 2;
    if (tid < 4 * n_iter) {
      ent = DR.table[(int64_t)it * nslots + 4 * r + (tid & 3)];
      ea = DR.table[(int64_t)it * nslots + 4 * r];                           // slot 0 describes the workgroup's A slice
    }
    __syncthreads();
    if (tid < 4 * n_iter) {
      const bool active = ent.x >= 0, gactive = ea.x >= 0;
      const DwRoleProblem& P = reinterpret_cast<const DwRoleProblem*>(probs)[active ? ent.x : 0];
      const DwRoleProblem& PA = reinterpret_cast<const DwRoleProblem*>(probs)[gactive ? ea.x : 0];
      const int la = ea.y / PA.tiles_n;
      const int tm = la % PA.tiles_m, z = la / PA.tiles_m;                   // (the same for every active slot of the workgroup)
      const int kbeg = ea.z * PA.kps;
      DwRec R;
      R.a = (unsigned long long)(PA.a + (int64_t)z * PA.a_sz);
      R.b = (unsigned long long)(P.b + (int64_t)z * P.b_sz);
      R.c = (unsigned long long)(P.c + (int64_t)z * P.c_sz);
      R.c2 = P.c2 ? (unsigned long long)(P.c2 + (int64_t)z * P.c_sz) : 0ull;
      R.a_bytes = ((PA.m - 1) + (PA.k - 1) * PA.a_sk + 1) * 4; R.a_sk = PA.a_sk; R.m = PA.m; R.m0 = tm * DWR_T;
      R.kbeg = kbeg; R.klen = gactive ? min(PA.k - kbeg, PA.kps) : 0;
      R.dept0 = gactive ? ((ea.w & 255) | (((ea.w >> 8) & 0xffff) << 8)) : DWR_DEP_NONE;
      R.w = (ent.w & 0x0fffffff) | (active ? DWR_ACTIVE : 0) | (gactive ? DWR_GACTIVE : 0);
      // row kbeg + rr of the chunk pairs with row kbeg + rr - b_shift of B (the recurrent product sum_t dA_t^T h_{t-1}: rows of
      // the first time step read zeros)
      R.b_bytes = ((max(P.n_valid, 1) - 1) + (max(P.k - P.b_shift, 1) - 1) * P.b_sk + 1) * 4; R.b_sk = P.b_sk;
      R.b_row0 = kbeg - P.b_shift; R.n0 = (ent.y % P.tiles_n) * DWR_T; R.n_valid = P.n_valid; R.ldc = P.ldc; R.alpha = P.alpha; R.pm = P.m;
      *reinterpret_cast<DwRec*>(recs + tid * 24) = R;
    }
    __syncthreads();
  }
  struct Item {                      // one record as this thread's slot sees it (every field wave-uniform, in scalar registers)
    bool active, gactive;
    unsigned long long a, b, c, c2;
    int a_bytes, a_sk, m, m0, kbeg, klen, dep, t0, w, b_bytes, b_sk, b_row0, n0, n_valid, ldc, pm;
    float alpha;
  };
  auto rfl = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  auto decode = [&](int it) {
    Item I;
    int v[24];
    if (it < n_iter) {
      const i32x4_t* rp = reinterpret_cast<const i32x4_t*>(recs + (4 * it + sub) * 24);
#pragma unroll
      for (int j = 0; j < 6; ++j) { const i32x4_t x = rp[j]; v[4 * j] = rfl(x[0]); v[4 * j + 1] = rfl(x[1]); v[4 * j + 2] = rfl(x[2]); v[4 * j + 3] = rfl(x[3]); }
    } else {
#pragma unroll
      for (int j = 0; j < 24; ++j) v[j] = 0;
    }
    auto u64 = [&](int i) { return (unsigned long long)(unsigned)v[i] | ((unsigned long long)(unsigned)v[i + 1] << 32); };
    I.a = u64(0); I.b = u64(2); I.c = u64(4); I.c2 = u64(6);
    I.a_bytes = v[8]; I.a_sk = v[9]; I.m = v[10]; I.m0 = v[11]; I.kbeg = v[12]; I.klen = v[13];
    I.dep = v[14] & 255; I.t0 = (v[14] >> 8) & 0xffff; I.w = v[15];
    I.b_bytes = v[16]; I.b_sk = v[17]; I.b_row0 = v[18]; I.n0 = v[19]; I.n_valid = v[20]; I.ldc = v[21];
    I.alpha = __builtin_bit_cast(float, v[22]); I.pm = v[23];
    I.active = (I.w & DWR_ACTIVE) != 0; I.gactive = (I.w & DWR_GACTIVE) != 0;
    return I;
  };
  // the B operand (batch columns, hidden states, records of the forward) never depends on this launch
  auto issue_b = [&](const Item& I, f32x4 (&rb)[BL]) {
    const __amdgpu_buffer_rsrc_t bres = __builtin_amdgcn_make_buffer_rsrc((void*)I.b, 0, I.b_bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < BL; ++j) {
      const int idx = t + j * 256;
      const int rr = idx >> 3, c4 = idx & 7;
      const int br = I.b_row0 + rr;
      const int offb = (I.active & (rr < I.klen) & (br >= 0) & (I.n0 + 4 * c4 < I.n_valid)) ? (br * I.b_sk + I.n0 + 4 * c4) * 4 : -16;
      rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bres, offb, 0, 0));
    }
  };
  // A may have been written inside this launch (dA, the latent gradients): agent-scope load (sc1), one per thread
  auto issue_a = [&](const Item& I) {
    const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc((void*)I.a, 0, I.a_bytes, 0x00020000);
    const int rr = tid >> 3, c4 = tid & 7;
    const int offa = (I.gactive & (rr < I.klen) & (I.m0 + 4 * c4 < I.m)) ? ((I.kbeg + rr) * I.a_sk + I.m0 + 4 * c4) * 4 : -16;
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ares, offa, 0, 16));
  };
  auto stamps_of = [&](const Item& I, int& n) -> const unsigned* {
    if (I.dep == DWR_DEP_LATENT) { n = 4 * DR.B; return DR.flags + 4 * DR.T * DWR_ROWS; }      // [4][B] dense
    n = DR.B;
    return DR.flags + ((I.dep - 1) * DR.T + I.t0) * DWR_ROWS;
  };

  Item cur = decode(0);
  f32x4 rb[BL], ra = {0.f, 0.f, 0.f, 0.f};
  issue_b(cur, rb);
  bool a_pref = false;
  if (cur.dep == DWR_DEP_NONE) { ra = issue_a(cur); a_pref = true; }
#pragma unroll 1
  for (int it = 0; it < n_iter; ++it) {
    LSTAMP(4, 16 + it);
    const Item nxt = decode(it + 1);
    // (1) the stamps of the current block, unless its A slice is already on its way; ONE wave per workgroup polls (many waves
    //     re-reading a few flag lines at the memory side wait on each other)
    if (!a_pref) {
      if (cur.dep != DWR_DEP_NONE && tid < 64) { int n; const unsigned* f = stamps_of(cur, n); dwr_wait(f, n, epoch, DR.ctl); }
      if (DR.any_dep) __syncthreads();
      ra = issue_a(cur);
    }
#if MFM_LAUNCH_STAMP
#define DWR_SUB(k) do { if (it == 5) LSTAMP(4, k); if (it == 1) LSTAMP(4, 6 + k); } while (0)
#else
#define DWR_SUB(k) ((void)0)
#endif
    DWR_SUB(1);
    // (2) are the next block's stamps there already?  (asked now, answered through LDS behind the barrier below)
    if (tid < 64) {
      bool ok = true;
      if (nxt.dep != DWR_DEP_NONE) { int n; const unsigned* f = stamps_of(nxt, n); ok = dwr_ready(f, n, epoch); }
      if (tid == 0) *nready_lds = ok ? 1 : 0;
    }
    DWR_SUB(2);
    // (3) operands of the current block -> LDS images
    {
      const int rr = tid >> 3, c4 = tid & 7;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ra[e] = (cur.m0 + 4 * c4 + e < cur.m) ? ra[e] : 0.0f;
        if (DR.bf16) ra[e] = (float)(__bf16)ra[e];
      }
      const int sw = (4 * c4 + 16 * (rr & 1)) & 31;
      *reinterpret_cast<f32x4*>(As + rr * DWR_T + sw) = ra;
    }
    if (cur.active) {
#pragma unroll
      for (int j = 0; j < BL; ++j) {
        const int idx = t + j * 256;
        const int rr = idx >> 3, c4 = idx & 7;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          rb[j][e] = (cur.n0 + 4 * c4 + e < cur.n_valid) ? rb[j][e] : 0.0f;
          if (DR.bf16) rb[j][e] = (float)(__bf16)rb[j][e];
        }
        const int sw = (4 * c4 + 16 * (rr & 1)) & 31;
        *reinterpret_cast<f32x4*>(Bs + rr * DWR_T + sw) = rb[j];
      }
    }
    __syncthreads();
    DWR_SUB(3);
    // (4) the next block's operands are requested before the current one is multiplied
    const bool nready = *nready_lds != 0;
    issue_b(nxt, rb);
    a_pref = false;
    if (nready) { ra = issue_a(nxt); a_pref = true; }
    DWR_SUB(4);
    // (5) product, epilogue
    if (cur.active) {
      const bool a1 = (cur.w & DWR_ACC1) != 0;
      f32x4 acc = a1 ? accs[1] : accs[0];
      if (cur.w & DWR_FIRST) acc = f32x4{0.f, 0.f, 0.f, 0.f};
      const int rot = 16 * (q & 1);
      const float* ap = As + q * DWR_T + ((16 * wm + bi + rot) & 31);
      const float* bp = Bs + q * DWR_T + ((16 * wn + bi + rot) & 31);
      const int nks = (cur.klen + 3) >> 2;           // rows klen .. 4 nks - 1 of the images are zeros (loaded out of range)
      int ks = 0;
      for (; ks + 4 <= nks; ks += 4) {
        float a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = ap[(ks + u) * 4 * DWR_T]; b[u] = bp[(ks + u) * 4 * DWR_T]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = mma16x16x4(a[u], b[u], acc);
      }
      for (; ks < nks; ++ks) acc = mma16x16x4(ap[ks * 4 * DWR_T], bp[ks * 4 * DWR_T], acc);
      if (a1) accs[1] = acc; else accs[0] = acc;
      if (cur.w & DWR_LAST) {
        float* __restrict__ C = reinterpret_cast<float*>(cur.c);
        float* __restrict__ C2 = reinterpret_cast<float*>(cur.c2);
        const int col = cur.n0 + 16 * wn + bi;
        if (col < cur.n_valid) {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int row = cur.m0 + 16 * wm + 4 * q + rr;
            if (row < cur.pm) {
              const float v = cur.alpha * acc[rr];
              if (cur.w & DWR_STORE) {             // the tile's only contribution, into a buffer that holds zeros: a plain store
                C[(int64_t)row * cur.ldc + col] = v;
                if (C2) C2[(int64_t)row * cur.ldc + col] = v;
              } else {
                atomicAdd(C + (int64_t)row * cur.ldc + col, v);
                if (C2) atomicAdd(C2 + (int64_t)row * cur.ldc + col, v);
              }
            }
          }
        }
      }
    }
    DWR_SUB(5);
    __syncthreads();          // the images (and the answer word) are free for the next block
    DWR_SUB(6);
    cur = nxt;
  }
}

}  // namespace mfm
