// Whole-sequence LSTM recurrence for SMALL batch tiles (latency regime): the kernels and launchers around the device bodies of
// lstm_seq_small_dev.h -- the plain multi-LSTM launches, the fold launches that also run the rows' latent chains (MFM_KL_EF), and
// the role-workgroup launches (input projections / weight gradients on the idle CUs, proj_role_dev.h / dw_role_dev.h).
#include "lstm_seq_small_dev.h"

namespace mfm {

#define MFM_SMALL_CASES(BODY)                                                                \
  switch (d.hk4) {                                                                           \
    case 2: BODY<2, 4>(d, L.T, L.B, tile, lds); break;                                          \
    case 4: BODY<4, 4>(d, L.T, L.B, tile, lds); break;                                          \
    case 6: BODY<6, 4>(d, L.T, L.B, tile, lds); break;                                          \
    case 8: BODY<8, 4>(d, L.T, L.B, tile, lds); break;                                          \
    case 10: BODY<10, 4>(d, L.T, L.B, tile, lds); break;                                        \
    case 12: BODY<12, 4>(d, L.T, L.B, tile, lds); break;                                        \
    case 14: BODY<14, 4>(d, L.T, L.B, tile, lds); break;                                        \
    case 16: BODY<16, 4>(d, L.T, L.B, tile, lds); break;                                        \
    case 18: BODY<18, 4>(d, L.T, L.B, tile, lds); break;                                        \
    case 20: BODY<20, 4>(d, L.T, L.B, tile, lds); break;                                        \
    case 22: BODY<22, 4>(d, L.T, L.B, tile, lds); break;                                        \
    case 24: BODY<24, 4>(d, L.T, L.B, tile, lds); break;                                        \
    case 26: BODY<26, 4>(d, L.T, L.B, tile, lds); break;                                        \
    case 28: BODY<28, 4>(d, L.T, L.B, tile, lds); break;                                        \
    case 30: BODY<30, 4>(d, L.T, L.B, tile, lds); break;                                        \
    case 32: BODY<32, 4>(d, L.T, L.B, tile, lds); break;                                        \
    default: break;                                                                          \
  }

template <bool BWD>
__global__ __launch_bounds__(1024) void lstm_seq_small_kernel(const SeqLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int di = 0;
  const int bid = blockIdx.x;
#pragma unroll 1
  for (int i = 1; i < L.count; ++i)
    if (bid >= L.d[i].block_begin) di = i;
  const SeqDev& d = L.d[di];
  const int tile = bid - d.block_begin;
  if (BWD) {
    MFM_SMALL_CASES(small_bwd_body)
  } else {
    MFM_SMALL_CASES(small_fwd_body)
  }
}

// Specialised launch for a fixed tuple of per-LSTM variants: with 4 bodies in the kernel instead of
// 16 the register allocator reaches single-variant quality (the 16-way switch spills ~80 VGPRs in the
// forward time loop, 4x slower).  The canonical MFM sizes (encoders 32/8/80/120, decoders 104/24/24)
// are pre-instantiated; any other size combination takes the generic kernel above.
// KS = 8 (backward, one-row tiles): 4 * Hp threads per workgroup, launched with at most 512 -> 256 VGPRs per thread for
// the doubled resident weights.
template <bool BWD, int R, int KS, int K0, int K1, int K2, int K3, int K4 = 0, int K5 = 0, bool BF = false>
__global__ __launch_bounds__(KS == 16 ? 1024 : 512) void lstm_seq_small_kernel4(const SeqLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int di = 0;
  const int bid = blockIdx.x;
  LSTAMP(BWD ? 3 : 1, 0);
  if constexpr (!BWD) {      // blocks behind the recurrence rows: this step's transposed-weight images (lstm_seq_dev.h)
    if (L.n_img > 0 && bid >= L.img_begin) { wt_img_write(L.img, L.n_img, bid - L.img_begin, L.n_img_blocks); LSTAMP(1, 15); return; }
  }
#pragma unroll 1
  for (int i = 1; i < L.count; ++i)
    if (bid >= L.d[i].block_begin) di = i;
  const SeqDev& d = L.d[di];
  const int tile = bid - d.block_begin;
#define MFM_ONE(IDX, KK)                                                         \
  if (KK > 0 && di == IDX) {                                                     \
    if (BWD) small_bwd_body<(KK > 0 ? KK : 2), R, KS>(d, L.T, L.B, tile, lds);   \
    else small_fwd_body<(KK > 0 ? KK : 2), R, false, BF>(d, L.T, L.B, tile, lds);           \
    LSTAMP(BWD ? 3 : 1, 15);                                                     \
    return;                                                                      \
  }
  MFM_ONE(0, K0) MFM_ONE(1, K1) MFM_ONE(2, K2) MFM_ONE(3, K3) MFM_ONE(4, K4) MFM_ONE(5, K5)
#undef MFM_ONE
}

template <int R, int KS, int K0, int K1, int K2, int K3, int K4 = 0, int K5 = 0>
static bool try_launch4(const SeqLaunch& L, bool bwd, int total, int threads, size_t lds_bytes, hipStream_t stream,
                        hipError_t* err) {
  const int want[6] = {K0, K1, K2, K3, K4, K5};
  int n = 0;
  for (int i = 0; i < 6; ++i) if (want[i] > 0) n = i + 1;
  if (L.count != n) return false;
  for (int i = 0; i < n; ++i) if (L.d[i].hk4 != want[i]) return false;
  *err = hipSuccess;
  if (lds_bytes > 64 * 1024) {
    *err = bwd ? hipFuncSetAttribute((const void*)lstm_seq_small_kernel4<true, R, KS, K0, K1, K2, K3, K4, K5>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)
               : hipFuncSetAttribute((const void*)lstm_seq_small_kernel4<false, R, KS, K0, K1, K2, K3, K4, K5>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (*err != hipSuccess) return true;
  }
  if (bwd)
    MFM_LAUNCH_TIMED((lstm_seq_small_kernel4<true, R, KS, K0, K1, K2, K3, K4, K5>), dim3(total), dim3(threads), lds_bytes, stream, L);
  else
    MFM_LAUNCH_TIMED((lstm_seq_small_kernel4<false, R, KS, K0, K1, K2, K3, K4, K5>), dim3(total), dim3(threads), lds_bytes, stream, L);
  return true;
}

template <int R>
static bool try_all(const SeqLaunch& L, bool bwd, int total, int threads, size_t lds_bytes, hipStream_t stream,
                    hipError_t* err) {
  // (from 2 rounds of workgroups on, the launcher orders the LSTMs of a call widest first: lstm_seq.hip)
  return try_launch4<R, 16, 8, 2, 20, 30>(L, bwd, total, threads, lds_bytes, stream, err) ||     // MFM_KL_EF encoders
         try_launch4<R, 16, 30, 20, 8, 2>(L, bwd, total, threads, lds_bytes, stream, err) ||
         try_launch4<R, 16, 26, 6, 6, 0>(L, bwd, total, threads, lds_bytes, stream, err) ||      // decoders
         try_launch4<R, 16, 30, 0, 0, 0>(L, bwd, total, threads, lds_bytes, stream, err) ||      // single-LSTM launches
         try_launch4<R, 16, 26, 0, 0, 0>(L, bwd, total, threads, lds_bytes, stream, err) ||
         try_launch4<R, 16, 8, 0, 0, 0>(L, bwd, total, threads, lds_bytes, stream, err) ||
         // (round 6: h = 24 / 8 / 80 alone.  scripts/bench_seq_group.py used to time these through the generic 4-row kernel -- the
         //  "h = 24 costs what h = 104 costs" of the round-5 review was that kernel, not the one-row bodies the plan runs)
         try_launch4<R, 16, 6, 0, 0, 0>(L, bwd, total, threads, lds_bytes, stream, err) ||
         try_launch4<R, 16, 2, 0, 0, 0>(L, bwd, total, threads, lds_bytes, stream, err) ||
         try_launch4<R, 16, 20, 0, 0, 0>(L, bwd, total, threads, lds_bytes, stream, err) ||
         // MFM / MFM_KL on the module path (mfm_model.py::seq_group): encoders 32/8/80 + MFN LSTM 88, MFN 64/48
         try_launch4<R, 16, 8, 2, 20, 22>(L, bwd, total, threads, lds_bytes, stream, err) ||
         try_launch4<R, 16, 16, 12, 0, 0>(L, bwd, total, threads, lds_bytes, stream, err) ||
         // MFM / MFM_KL on the fused plan: all six LSTMs of the step (encoders 32/8/80, MFN 88/64/48) in one launch
         try_launch4<R, 16, 8, 2, 20, 22, 16, 12>(L, bwd, total, threads, lds_bytes, stream, err);
}

// backward, one-row tiles, 8 k-slices (4 * Hp threads): the pre-instantiated size tuples
static bool try_all_fat(const SeqLaunch& L, int total, int threads, size_t lds_bytes, hipStream_t stream, hipError_t* err) {
  return try_launch4<1, 8, 8, 2, 20, 30>(L, true, total, threads, lds_bytes, stream, err) ||
         try_launch4<1, 8, 26, 6, 6, 0>(L, true, total, threads, lds_bytes, stream, err) ||
         try_launch4<1, 8, 30, 0, 0, 0>(L, true, total, threads, lds_bytes, stream, err) ||
         try_launch4<1, 8, 26, 0, 0, 0>(L, true, total, threads, lds_bytes, stream, err) ||
         try_launch4<1, 8, 8, 0, 0, 0>(L, true, total, threads, lds_bytes, stream, err) ||
         try_launch4<1, 8, 8, 2, 20, 22>(L, true, total, threads, lds_bytes, stream, err) ||
         try_launch4<1, 8, 16, 12, 0, 0>(L, true, total, threads, lds_bytes, stream, err) ||
         try_launch4<1, 8, 8, 2, 20, 22, 16, 12>(L, true, total, threads, lds_bytes, stream, err);
}

// Fold launch (MFM_KL_EF at small batches): the workgroup of (encoder e, batch row b) also runs modality chain e of row
// b's latent stack -- forward right behind its last time step, backward ahead of its BPTT.  The four chains are
// independent inside the stack and map one to one onto the four encoders, so nothing is exchanged between workgroups:
// what used to be a launch boundary (~3 us) plus the chain kernel's cold prologue becomes a barrier.  One-row tiles
// and the canonical size tuple only; anything else keeps the separate launches.
template <bool BWD, int K0, int K1, int K2, int K3>
__global__ __launch_bounds__(1024) void lstm_seq_small_fold_kernel(const SeqLaunch L, const LatentDev LD, const float* __restrict__ params,
                                                                   float* __restrict__ grads) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int di = 0;
  const int bid = blockIdx.x;
  if constexpr (!BWD) {      // blocks behind the recurrence rows: this step's transposed-weight images (lstm_seq_dev.h)
    if (L.n_img > 0 && bid >= L.img_begin) { wt_img_write(L.img, L.n_img, bid - L.img_begin, L.n_img_blocks); return; }
  }
#pragma unroll 1
  for (int i = 1; i < L.count; ++i)
    if (bid >= L.d[i].block_begin) di = i;
  const SeqDev& d = L.d[di];
  const int tile = bid - d.block_begin;          // == batch row (one-row tiles)
  if constexpr (BWD) {
    latent_bwd_row_body<false>(LD, params, grads, tile, di, lds);
    sync_stores();           // d h_T of this (row, encoder) is in memory (and the LDS is free) before the BPTT reads it
  }
#define MFM_ONE(IDX, KK)                                                        \
  if (KK > 0 && di == IDX) {                                                    \
    if (BWD) small_bwd_body<(KK > 0 ? KK : 2), 1, 16>(d, L.T, L.B, tile, lds);  \
    else small_fwd_body<(KK > 0 ? KK : 2), 1>(d, L.T, L.B, tile, lds);          \
  }
  MFM_ONE(0, K0) MFM_ONE(1, K1) MFM_ONE(2, K2) MFM_ONE(3, K3)
#undef MFM_ONE
  if constexpr (!BWD) {
    sync_stores();           // every store of the last time step has been acknowledged: h_T of this row is readable
    latent_fwd_row_body<false>(LD, params, tile, di, lds, true);
  }
}

// Forward fold launch with projection role workgroups (proj_role_dev.h): blocks [0, n_role) produce the x-projections
// (and clear the launch's zero spans), blocks [n_role, n_role + 4 B) are the fold launch's (encoder, row) workgroups.
template <int K0, int K1, int K2, int K3>
__global__ __launch_bounds__(1024) void lstm_seq_small_foldproj_kernel(const SeqLaunch L, const LatentDev LD, const ProjRole PR,
                                                                       const float* __restrict__ params) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int bid = blockIdx.x;
  const unsigned epoch = ho_epoch(PR.epoch, PR.tick);
  LSTAMP(0, 0);
  if (bid < PR.n_role) { proj_role_body(L, PR, epoch, lds); LSTAMP(0, 15); return; }
  int di = 0;
#pragma unroll 1
  for (int i = 1; i < L.count; ++i)
    if (bid >= L.d[i].block_begin) di = i;
  const SeqDev& d = L.d[di];
  const int tile = bid - d.block_begin;
  const unsigned* flg = PR.flags + (int64_t)di * L.T * PROJ_ROLE_FLAGS;
  const int ncb = PR.e[di].ncb;
  // PR.lat_pre (every encoder loads its weights straight into registers: no LDS weight panel): the chain's tables go to LDS NOW,
  // while the row waits for its first projections anyway; the recurrence's buffers lie behind the chain's region, and the chain
  // takes h_T from the recurrence's LDS buffer instead of reading it back from memory behind a store wait
  const bool pre = PR.lat_pre != 0;
  float* seq_lds = lds;
  if (pre) {
    latent_fwd_row_body<false>(LD, params, tile, di, lds, true, 1);
    seq_lds = lds + latent_fwd_lds_floats(LD.rec_size);
  }
  const float* h_lds = nullptr;
#define MFM_ONE(IDX, KK) \
  if (KK > 0 && di == IDX) h_lds = small_fwd_body<(KK > 0 ? KK : 2), 1, true>(d, L.T, L.B, tile, seq_lds, flg, epoch, ncb, PR.ctl);
  MFM_ONE(0, K0) MFM_ONE(1, K1) MFM_ONE(2, K2) MFM_ONE(3, K3)
#undef MFM_ONE
  // the loss slots are cleared by the producers of t = 0: all of them have passed before this row's chain adds to them
  // (the four flag sets are requested together: one round trip)
  if (threadIdx.x < 64) {
    unsigned fv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) fv[e] = proj_flags_load(PR.flags + (int64_t)e * L.T * PROJ_ROLE_FLAGS, PR.e[e].ncb);
#pragma unroll
    for (int e = 0; e < 4; ++e) proj_flags_wait(PR.flags + (int64_t)e * L.T * PROJ_ROLE_FLAGS, fv[e], epoch, PR.e[e].ncb, PR.ctl);
  }
  LSTAMP(0, 9);
  if (pre) {
    LSTAMP(0, 8);
    // (its first barrier orders the flag wait above)
    latent_fwd_row_body<false>(LD, params, tile, di, lds, true, 2, h_lds, PR.lat_tail ? LD.tail_from : MFM_LAT_MAXSTAGES);
  } else {
    sync_stores();           // every store of the last time step has been acknowledged: h_T of this row is readable
    LSTAMP(0, 8);
    latent_fwd_row_body<false>(LD, params, tile, di, lds, true);
  }
  LSTAMP_W(0, 15);
}

// Backward fold launch with weight-gradient role workgroups (dw_role_dev.h): blocks [0, 4 B) are the fold launch's
// (encoder, row) workgroups, blocks [4 B, 4 B + n_role) run the table of weight-gradient blocks.
template <int K0, int K1, int K2, int K3>
__global__ __launch_bounds__(1024) void lstm_seq_small_folddw_kernel(const SeqLaunch L, const LatentDev LD, const DwRole DR,
                                                                     const float* __restrict__ params, float* __restrict__ grads) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int bid = blockIdx.x;
  const unsigned epoch = ho_epoch(DR.epoch, DR.tick);
  LSTAMP(4, 0);
  if (bid >= 4 * L.B) {
    // (the DwRole descriptor as it lies in the kernel-argument segment, behind L and LD: dw_role_dev.h reads it with per-lane indices)
    typedef __attribute__((address_space(4))) const char kchar;
    const char* kargs = (const char*)(kchar*)__builtin_amdgcn_kernarg_segment_ptr();
    dw_role_body(DR, epoch, lds, reinterpret_cast<const int*>(kargs + sizeof(SeqLaunch) + sizeof(LatentDev)));
    LSTAMP_W(4, 15);
    return;
  }
  // a plan whose status word is set (a hand-over of this or of an earlier step gave up: the forward's projections may be
  // garbage) must not train: the guard word of the gradient buffer makes the optimizer skip (and travels through the all-reduce)
  if (bid == 0 && threadIdx.x == 0 && DR.ctl.status && DR.ctl.poison && *DR.ctl.status != 0u)
    __hip_atomic_store(DR.ctl.poison, __builtin_nanf(""), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int di = 0;
#pragma unroll 1
  for (int i = 1; i < L.count; ++i)
    if (bid >= L.d[i].block_begin) di = i;
  const SeqDev& d = L.d[di];
  const int tile = bid - d.block_begin;          // == batch row (one-row tiles)
  unsigned* stamp = DR.flags + (int64_t)di * L.T * DWR_ROWS + tile;
  unsigned* lat_stamp = DR.flags + 4 * L.T * DWR_ROWS + di * L.B + tile;
  const bool fault = DR.fault != 0 && bid == 0;
  if (DR.lat_split) {
    // the chain up to its last stage; the BPTT takes d h_T from the chain's gradient record in LDS (which lies behind everything the
    // BPTT keeps in LDS: seq_small_folddw_launch), and the chain's stores / atomics go out while the BPTT's weights travel
    latent_bwd_row_body<false>(LD, params, grads, tile, di, lds, 1);        // (ends behind a barrier)
    LSTAMP(4, 9);
    const float* dh_lds = lds + latent_bwd_grd_floats(LD.rec_size) + LD.in_off[di];
#define MFM_ONE(IDX, KK) \
    if (KK > 0 && di == IDX) small_bwd_body_x<(KK > 0 ? KK : 2), 1, 16, true, LatentDev>(d, L.T, L.B, tile, lds, stamp, epoch, fault, LD, params, grads, di, dh_lds, lat_stamp);
    MFM_ONE(0, K0) MFM_ONE(1, K1) MFM_ONE(2, K2) MFM_ONE(3, K3)
#undef MFM_ONE
  } else {
    latent_bwd_row_body<false>(LD, params, grads, tile, di, lds);
    LSTAMP(4, 9);
    sync_stores();           // d h_T of this (row, encoder) is in memory (and the LDS is free) before the BPTT reads it
    LSTAMP(4, 8);
    if (threadIdx.x == 0) dwr_stamp(lat_stamp, epoch);
#define MFM_ONE(IDX, KK) \
    if (KK > 0 && di == IDX) small_bwd_body<(KK > 0 ? KK : 2), 1, 16, true>(d, L.T, L.B, tile, lds, stamp, epoch, fault);
    MFM_ONE(0, K0) MFM_ONE(1, K1) MFM_ONE(2, K2) MFM_ONE(3, K3)
#undef MFM_ONE
  }
  LSTAMP_W(4, 15);
}

// Decoder recurrences with TAIL blocks (round 6): blocks [0, 3 B) are kernel4<false>'s rows of the three decoders; blocks
// [tail_begin, tail_begin + 4 B) finish the latent forward chains the encoder launch left behind the z -> f MLPs -- classifier,
// logvar heads, KLD and discriminative loss, y_hat, the rest of the saved record (latent_row_dev.h, mode 3).  Nothing in this
// launch waits for them; they run on CUs the 96 decoder rows leave idle.
template <int K0, int K1, int K2>
__global__ __launch_bounds__(1024) void lstm_seq_small_dectail_kernel(const SeqLaunch L, const LatentDev LD, const float* __restrict__ params,
                                                                      const int tail_begin) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int bid = blockIdx.x;
  LSTAMP(1, 0);
  if (bid >= tail_begin) {
    const int r = bid - tail_begin;
    latent_fwd_row_body<false>(LD, params, r % L.B, r / L.B, lds, true, 3);
    LSTAMP(1, 15);
    return;
  }
  int di = 0;
#pragma unroll 1
  for (int i = 1; i < L.count; ++i)
    if (bid >= L.d[i].block_begin) di = i;
  const SeqDev& d = L.d[di];
  const int tile = bid - d.block_begin;
#define MFM_ONE(IDX, KK) \
  if (KK > 0 && di == IDX) small_fwd_body<(KK > 0 ? KK : 2), 1>(d, L.T, L.B, tile, lds);
  MFM_ONE(0, K0) MFM_ONE(1, K1) MFM_ONE(2, K2)
#undef MFM_ONE
  LSTAMP(1, 15);
}

// Decoder BPTTs with HEAD blocks (round 6): blocks [0, 3 B) are kernel4<true>'s rows; blocks [head_begin, head_begin + 4 B) run the
// part of the latent BACKWARD chains that does not wait for the decoders (latent_row_dev.h, mode 3) on CUs the 96 rows leave idle.
template <int K0, int K1, int K2>
__global__ __launch_bounds__(1024) void lstm_seq_small_decbwd_head_kernel(const SeqLaunch L, const LatentDev LD, const float* __restrict__ params,
                                                                          float* __restrict__ grads, const int head_begin) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int bid = blockIdx.x;
  LSTAMP(3, 0);
  if (bid >= head_begin) {
    const int r = bid - head_begin;
    latent_bwd_row_body<false>(LD, params, grads, r % L.B, r / L.B, lds, 3);
    LSTAMP(3, 15);
    return;
  }
  int di = 0;
#pragma unroll 1
  for (int i = 1; i < L.count; ++i)
    if (bid >= L.d[i].block_begin) di = i;
  const SeqDev& d = L.d[di];
  const int tile = bid - d.block_begin;
#define MFM_ONE(IDX, KK) \
  if (KK > 0 && di == IDX) small_bwd_body<(KK > 0 ? KK : 2), 1, 16>(d, L.T, L.B, tile, lds);
  MFM_ONE(0, K0) MFM_ONE(1, K1) MFM_ONE(2, K2)
#undef MFM_ONE
  LSTAMP(3, 15);
}

// `no_panel` (forward, one-row tiles, every h % 4 == 0): the weights go straight into registers, the panel is not needed
static size_t small_lds_bytes(const SeqLaunch& L, bool bwd, int R, bool no_panel = false) {
  size_t lds_bytes = 0;
  for (int i = 0; i < L.count; ++i) {
    const SeqDev& d = L.d[i];
    const size_t HK = (size_t)d.hk4 * 4;
    const size_t HKB = (HK + 15) / 16 * 16;
    const size_t hh = no_panel ? 0 : (size_t)d.h * d.h;
    // forward: h ring + max(weight panel, output record + x-projection record); backward: dA ring +
    // saved-activation record + weight panel (see the bodies)
    const size_t rec = (2 * 6 + 2 * 4) * HKB * R;
    // backward, one-row tiles: the weight panel holds two gates (stage_gates2)
    const size_t need = (bwd ? (2 * 4 + 2 * 7) * HKB * R + (R == 1 ? 2 : 1) * hh : 2 * HKB * R + (2 * hh > rec ? 2 * hh : rec)) * sizeof(float);
    if (need > lds_bytes) lds_bytes = need;
  }
  return (lds_bytes + 15) / 16 * 16;
}

int seq_small_launch(SeqLaunch& L, bool bwd, hipStream_t stream) {
  LSTAMP_BIND();
  int max_threads = 64;
  for (int i = 0; i < L.count; ++i)
    if (8 * L.d[i].Hp > max_threads) max_threads = 8 * L.d[i].Hp;
  // Row-tile size.  Fewer rows per workgroup = shorter serial chain per time step (the step time is
  // VALU-issue bound: ~60 FMAs per thread and row), and that outweighs running several rounds of
  // workgroups per CU for as long as measured: 1-row tiles win up to 6 workgroups per CU (B=32..384 with four
  // LSTMs per launch: +12 % per step at B=128/256 over 2- and 4-row tiles), 4-row tiles from there on
  // (B=512: 0.856 vs 0.948 ms); 2-row tiles never did (profiles/r01l_rows_per_workgroup.txt).  Only the
  // pre-instantiated size tuples have R < 4 variants.
  const int cus = device_cus();
  int R = ((long)L.count * L.B < 6L * cus) ? 1 : 4;
  if (const char* e = opt_get("MFM_SEQ_ROWS")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) R = v; }
  for (int attempt = 0; attempt < 2; ++attempt) {
    const int tiles = cdiv(L.B, R);
    int total = 0;
    for (int i = 0; i < L.count; ++i) { L.d[i].block_begin = total; total += tiles; }
    const size_t lds_bytes = small_lds_bytes(L, bwd, R);
    hipError_t err = hipSuccess;
    bool done = false;
    // forward, one-row tiles, idle CUs left: a few more blocks write this step's transposed-weight images
    const int rows_total = total;
    L.n_img_blocks = 0;
    // (no idle CU left: the writers still pay -- they run in the launch's last round, ~2 us, against ~9 us of staging saved;
    //  MFM_WT_IMG_FULL=0: only with idle CUs)
    const bool img_full = !(opt_get("MFM_WT_IMG_FULL") && atoi(opt_get("MFM_WT_IMG_FULL")) == 0);
    if (!bwd && R == 1 && L.n_img > 0 && (cus - total >= 8 || img_full)) {
      L.img_begin = total; L.n_img_blocks = cus - total >= 8 ? std::min(cus - total, 64) : 32; total += L.n_img_blocks;
    }
    // backward, one-row tiles: MFM_SEQ_KS=8 selects 8 k-slices per unit pair (half the threads, twice the FMAs each).
    // Opt-in: measured equal (encoders 30.5 vs 29.8 us) or slower (decoders 37.9 vs 32.6 us, 24 spilled registers) at B=32
    // (profiles/r02_seq_ks8.txt) -- the step is a latency chain (LDS hand-over, barrier, reduction), not issue-bound enough
    // for fewer, fatter waves to pay
    bool fat = false;
    if (const char* e = opt_get("MFM_SEQ_KS")) fat = bwd && R == 1 && max_threads / 2 <= 512 && atoi(e) == 8;
    if (fat) done = try_all_fat(L, total, max_threads / 2, lds_bytes, stream, &err);
    // bf16 plans, forward, one-row tiles: the decoders' recurrent product on v_dot2c_f32_bf16 (opt-in, small_fwd_body<.., BF>)
    if (!done && !bwd && R == 1 && L.bf16_dot && L.count == 3 && L.d[0].hk4 == 26 && L.d[1].hk4 == 6 && L.d[2].hk4 == 6) {
      if (lds_bytes > 64 * 1024)
        err = hipFuncSetAttribute((const void*)lstm_seq_small_kernel4<false, 1, 16, 26, 6, 6, 0, 0, 0, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      if (err == hipSuccess)
        MFM_LAUNCH_TIMED((lstm_seq_small_kernel4<false, 1, 16, 26, 6, 6, 0, 0, 0, true>), dim3(total), dim3(max_threads), lds_bytes, stream, L);
      done = true;
    }
    if (done) {}
    else if (R == 1) done = try_all<1>(L, bwd, total, max_threads, lds_bytes, stream, &err);
    else if (R == 2) done = try_all<2>(L, bwd, total, max_threads, lds_bytes, stream, &err);
    else done = try_all<4>(L, bwd, total, max_threads, lds_bytes, stream, &err);
    if (done) {
      if (err != hipSuccess) return hip_fail(err, "hipFuncSetAttribute(lstm_seq_small_kernel4)");
      MFM_LAUNCH_CHECK(bwd ? "lstm_seq_small_bwd_kernel4" : "lstm_seq_small_fwd_kernel4");
      if (L.n_img_blocks == 0) L.n_img = 0;        // (tells the caller that no images were written)
      return MFM_OK;
    }
    (void)rows_total;
    R = 4;    // no specialised kernel for this size tuple: generic 16-variant kernel, 4-row tiles
  }
  L.n_img = 0; L.n_img_blocks = 0;
  const int tiles = cdiv(L.B, 4);
  int total = 0;
  for (int i = 0; i < L.count; ++i) { L.d[i].block_begin = total; total += tiles; }
  const size_t lds_bytes = small_lds_bytes(L, bwd, 4);
  if (lds_bytes > 64 * 1024) {
    MFM_HIP_CHECK(hipFuncSetAttribute((const void*)lstm_seq_small_kernel<true>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    MFM_HIP_CHECK(hipFuncSetAttribute((const void*)lstm_seq_small_kernel<false>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  }
  if (bwd)
    MFM_LAUNCH_TIMED(lstm_seq_small_kernel<true>, dim3(total), dim3(max_threads), lds_bytes, stream, L);
  else
    MFM_LAUNCH_TIMED(lstm_seq_small_kernel<false>, dim3(total), dim3(max_threads), lds_bytes, stream, L);
  MFM_LAUNCH_CHECK(bwd ? "lstm_seq_small_bwd_kernel" : "lstm_seq_small_fwd_kernel");
  return MFM_OK;
}


// MFM_OK: launched.  MFM_ERR_UNSUPPORTED: not a case the fold kernels take (the caller issues the separate launches).
// Whether the plan may use the role-workgroup launches at all is the plan's "handover" option (plan.hip): the in-launch
// hand-overs count on this launch becoming resident workgroup by workgroup in block order; launches of two queues that
// together exceed the CUs can block each other's producers until the waits give up (status word, poisoned gradient guard).
bool seq_small_folddw_supported(int T, int B) {
  if (const char* e = opt_get("MFM_DW_FOLD")) { if (atoi(e) == 0) return false; }
  if (opt_get("MFM_SEQ_KS") || opt_get("MFM_SEQ_ROWS")) return false;
  if (const char* e = opt_get("MFM_LATENT_FOLD")) { if (atoi(e) == 0) return false; }
  return T >= 1 && B >= 1 && B <= DWR_ROWS && device_cus() - 4 * B >= 32;
}

int seq_small_folddw_launch(SeqLaunch& L, const LatentDev& LD, DwRole& DR, const float* params, float* grads, hipStream_t stream) {
  const int want[4] = {8, 2, 20, 30};
  if (L.count != 4 || !LD.row_path || LD.nch != 4 || LD.pre || LD.B != L.B) return MFM_ERR_UNSUPPORTED;
  for (int i = 0; i < 4; ++i)
    if (L.d[i].hk4 != want[i] || L.d[i].is_dec) return MFM_ERR_UNSUPPORTED;
  if (!seq_small_folddw_supported(L.T, L.B)) return MFM_ERR_UNSUPPORTED;
  LSTAMP_BIND();
  const int n_role = DR.n_role;          // the block table was laid out for this many role workgroups
  if (n_role < 1 || n_role > device_cus() - 4 * L.B) return MFM_ERR_UNSUPPORTED;
  DR.T = L.T; DR.B = L.B;
  int total = 0;
  for (int i = 0; i < 4; ++i) { L.d[i].block_begin = total; total += L.B; }
  total += n_role;
  size_t lds_bytes = small_lds_bytes(L, true, 1);
  const size_t lat = ((size_t)MFM_LAT_MAXSTAGES * MFM_LAT_ROW_THREADS * 4 + 2 * (size_t)LD.rec_size) * sizeof(float);
  const size_t role = (size_t)DWR_LDS_FLOATS * sizeof(float);      // one shared A image + four B images + the pipeline's answer word + the block records
  if (DR.n_iter * 4 > DWR_MAXREC) return MFM_ERR_UNSUPPORTED;
  // the chain's gradient record must survive into the BPTT's prologue: it has to lie behind everything the BPTT keeps in LDS, and
  // every encoder must take its transposed weights from this step's images (no staging panel)
  DR.lat_split = 0;
  if (!(opt_get("MFM_LATENT_SPLIT") && atoi(opt_get("MFM_LATENT_SPLIT")) == 0)) {
    bool imgs = true;
    for (int i = 0; i < 4; ++i) imgs = imgs && L.d[i].wt_img != nullptr;
    if (imgs && lds_bytes <= (size_t)latent_bwd_grd_floats(LD.rec_size) * sizeof(float)) DR.lat_split = 1;
  }
  lds_bytes = std::max(lds_bytes, std::max(lat, role));
  if (lds_bytes > 160 * 1024) return MFM_ERR_UNSUPPORTED;
  MFM_HIP_CHECK(hipFuncSetAttribute((const void*)lstm_seq_small_folddw_kernel<8, 2, 20, 30>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  MFM_LAUNCH_TIMED((lstm_seq_small_folddw_kernel<8, 2, 20, 30>), dim3(total), dim3(1024), lds_bytes, stream, L, LD, DR, params, grads);
  MFM_LAUNCH_CHECK("lstm_seq_small_folddw_kernel");
  return MFM_OK;
}

// Can the decoder launch of this plan carry the tail blocks?  (asked by the encoder launch before it leaves the tails behind)
bool seq_small_dectail_supported(int T, int B, const int* dec_h, const LatentDev& LD) {
  if (const char* e = opt_get("MFM_LATENT_TAIL")) { if (atoi(e) == 0) return false; }
  if (opt_get("MFM_SEQ_KS") || opt_get("MFM_SEQ_ROWS") || opt_get("MFM_BF16_DOT")) return false;
  const int want[3] = {26, 6, 6};
  for (int i = 0; i < 3; ++i)
    if (round_up(cdiv(dec_h[i], 4), 2) != want[i]) return false;
  if (!LD.row_path || LD.nch != 4 || LD.pre || LD.B != B || LD.tail_from < 1 || LD.tail_from >= LD.nstages) return false;
  return T >= 1 && 3 * B + 4 * B <= device_cus();
}

// MFM_OK: launched (decoder rows + tail blocks).  MFM_ERR_UNSUPPORTED: the caller runs the tails as a launch of their own.
int seq_small_dectail_launch(SeqLaunch& L, const LatentDev& LD, const float* params, hipStream_t stream) {
  LSTAMP_BIND();
  const int want[3] = {26, 6, 6};
  if (L.count != 3 || L.bf16_dot) return MFM_ERR_UNSUPPORTED;
  int hh[3];
  for (int i = 0; i < 3; ++i) {
    if (L.d[i].hk4 != want[i] || !L.d[i].is_dec) return MFM_ERR_UNSUPPORTED;
    hh[i] = L.d[i].h;
  }
  if (!seq_small_dectail_supported(L.T, L.B, hh, LD)) return MFM_ERR_UNSUPPORTED;
  int total = 0, max_threads = 1024;          // (the chain's work items are tabulated for 1024 threads)
  bool direct = true;
  for (int i = 0; i < 3; ++i) { L.d[i].block_begin = total; total += L.B; direct = direct && (L.d[i].h & 3) == 0 && L.d[i].wf_img && L.d[i].wf1_img; }
  const int tail_begin = total;
  total += 4 * L.B;
  L.n_img = 0; L.n_img_blocks = 0;
  const size_t lat = (size_t)latent_fwd_lds_floats(LD.rec_size) * sizeof(float);
  const size_t lds_bytes = std::max(lat, small_lds_bytes(L, false, 1, direct));
  if (lds_bytes > 160 * 1024) return MFM_ERR_UNSUPPORTED;
  MFM_HIP_CHECK(hipFuncSetAttribute((const void*)lstm_seq_small_dectail_kernel<26, 6, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  MFM_LAUNCH_TIMED((lstm_seq_small_dectail_kernel<26, 6, 6>), dim3(total), dim3(max_threads), lds_bytes, stream, L, LD, params, tail_begin);
  MFM_LAUNCH_CHECK("lstm_seq_small_dectail_kernel");
  return MFM_OK;
}

// MFM_OK: launched (decoder BPTT rows + head blocks).  MFM_ERR_UNSUPPORTED: the caller runs the plain BPTT launch, the encoder
// launch its whole chain.
int seq_small_decbwd_head_launch(SeqLaunch& L, const LatentDev& LD, const float* params, float* grads, hipStream_t stream) {
  LSTAMP_BIND();
  if (const char* e = opt_get("MFM_LATENT_BWD_HEAD")) { if (atoi(e) == 0) return MFM_ERR_UNSUPPORTED; }
  const int want[3] = {26, 6, 6};
  if (L.count != 3) return MFM_ERR_UNSUPPORTED;
  int hh[3];
  for (int i = 0; i < 3; ++i) {
    if (L.d[i].hk4 != want[i] || !L.d[i].is_dec || !L.d[i].wt_img) return MFM_ERR_UNSUPPORTED;
    hh[i] = L.d[i].h;
  }
  if (!seq_small_dectail_supported(L.T, L.B, hh, LD) || !LD.grd_out) return MFM_ERR_UNSUPPORTED;
  int total = 0;
  for (int i = 0; i < 3; ++i) { L.d[i].block_begin = total; total += L.B; }
  const int head_begin = total;
  total += 4 * L.B;
  const size_t lat = ((size_t)MFM_LAT_MAXSTAGES * MFM_LAT_ROW_THREADS * 4 + 2 * (size_t)LD.rec_size) * sizeof(float);
  const size_t lds_bytes = std::max(lat, small_lds_bytes(L, true, 1));
  if (lds_bytes > 160 * 1024) return MFM_ERR_UNSUPPORTED;
  MFM_HIP_CHECK(hipFuncSetAttribute((const void*)lstm_seq_small_decbwd_head_kernel<26, 6, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  MFM_LAUNCH_TIMED((lstm_seq_small_decbwd_head_kernel<26, 6, 6>), dim3(total), dim3(1024), lds_bytes, stream, L, LD, params, grads, head_begin);
  MFM_LAUNCH_CHECK("lstm_seq_small_decbwd_head_kernel");
  return MFM_OK;
}

// Projection role workgroups for a forward fold launch: shapes the role kernel takes and enough idle CUs
bool seq_small_foldproj_supported(int T, int B, const int* h, const int* k, int n_enc) {
  if (const char* e = opt_get("MFM_PROJ_FOLD")) { if (atoi(e) == 0) return false; }
  if (opt_get("MFM_SEQ_KS") || opt_get("MFM_SEQ_ROWS")) return false;
  if (const char* e = opt_get("MFM_LATENT_FOLD")) { if (atoi(e) == 0) return false; }
  if (n_enc != 4 || T < 1 || B < 1 || B > PROJ_ROLE_CB) return false;
  if (device_cus() - 4 * B < PROJ_ROLE_SLOTS) return false;
  int slots = 0, kmax = 0;
  for (int e = 0; e < 4; ++e) {
    if (k[e] < 1 || k[e] > 128 * PROJ_ROLE_MAXG) return false;
    slots += 4 * round_up(h[e], 16) / PROJ_ROLE_CB;
    if (4 * round_up(h[e], 16) / PROJ_ROLE_CB > PROJ_ROLE_FLAGS) return false;
    kmax = std::max(kmax, k[e]);
  }
  if (slots > PROJ_ROLE_SLOTS) return false;
  const int ks = proj_role_kstride(kmax);
  return ((size_t)2 * PROJ_ROLE_CB * ks + 16 * PROJ_ROLE_PW) * sizeof(float) <= 160 * 1024;
}

int seq_small_foldproj_launch(SeqLaunch& L, const LatentDev& LD, ProjRole& PR, const float* params, hipStream_t stream) {
  const int want[4] = {8, 2, 20, 30};
  if (L.count != 4 || !LD.row_path || LD.nch != 4 || LD.pre || LD.B != L.B) return MFM_ERR_UNSUPPORTED;
  for (int i = 0; i < 4; ++i)
    if (L.d[i].hk4 != want[i] || L.d[i].is_dec) return MFM_ERR_UNSUPPORTED;
  int hh[4], kk[4];
  for (int i = 0; i < 4; ++i) { hh[i] = L.d[i].h; kk[i] = PR.e[i].k; }
  if (!seq_small_foldproj_supported(L.T, L.B, hh, kk, 4)) return MFM_ERR_UNSUPPORTED;
  LSTAMP_BIND();
  int n_role = (device_cus() - 4 * L.B) / PROJ_ROLE_SLOTS * PROJ_ROLE_SLOTS;
  if (const char* e = opt_get("MFM_PROJ_FOLD_ROLES")) { const int v = atoi(e) / PROJ_ROLE_SLOTS * PROJ_ROLE_SLOTS; if (v >= PROJ_ROLE_SLOTS) n_role = v; }
  PR.n_role = n_role; PR.groups = n_role / PROJ_ROLE_SLOTS;
  int kmax = 0, cb = 0;
  for (int i = 0; i < 4; ++i) {
    kmax = std::max(kmax, PR.e[i].k);
    PR.e[i].cb_begin = cb; PR.e[i].ncb = 4 * L.d[i].Hp / PROJ_ROLE_CB;
    cb += PR.e[i].ncb;
  }
  PR.kstride = proj_role_kstride(kmax);
  int total = n_role;
  for (int i = 0; i < 4; ++i) { L.d[i].block_begin = total; total += L.B; }
  size_t lds_bytes = small_lds_bytes(L, false, 1);
  const size_t lat = ((size_t)MFM_LAT_MAXSTAGES * MFM_LAT_ROW_THREADS * 4 + (size_t)LD.rec_size) * sizeof(float);
  const size_t role = ((size_t)2 * PROJ_ROLE_CB * PR.kstride + 16 * PROJ_ROLE_PW) * sizeof(float);
  lds_bytes = std::max(lds_bytes, std::max(lat, role));
  // chain tables preloaded in front of the time loop: chain region + recurrence buffers side by side (lstm_seq_small_foldproj_kernel)
  PR.lat_pre = 0;
  if (!(opt_get("MFM_LATENT_PRELOAD") && atoi(opt_get("MFM_LATENT_PRELOAD")) == 0)) {
    bool direct = true;
    for (int i = 0; i < 4; ++i) direct = direct && (L.d[i].h & 3) == 0;
    const size_t side = (size_t)latent_fwd_lds_floats(LD.rec_size) * sizeof(float) + small_lds_bytes(L, false, 1, true);
    if (direct && side <= 160 * 1024) { PR.lat_pre = 1; lds_bytes = std::max(side, role); }
  }
  if (!PR.lat_pre) PR.lat_tail = 0;          // (the head / tail split is a form of the preloaded chain)
  if (lds_bytes > 160 * 1024) return MFM_ERR_UNSUPPORTED;
  MFM_HIP_CHECK(hipFuncSetAttribute((const void*)lstm_seq_small_foldproj_kernel<8, 2, 20, 30>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  MFM_LAUNCH_TIMED((lstm_seq_small_foldproj_kernel<8, 2, 20, 30>), dim3(total), dim3(1024), lds_bytes, stream, L, LD, PR, params);
  MFM_LAUNCH_CHECK("lstm_seq_small_foldproj_kernel");
  return MFM_OK;
}

int seq_small_fold_launch(SeqLaunch& L, bool bwd, const LatentDev& LD, const float* params, float* grads, hipStream_t stream) {
  if (const char* e = opt_get("MFM_LATENT_FOLD")) { if (atoi(e) == 0) return MFM_ERR_UNSUPPORTED; }
  if (opt_get("MFM_SEQ_KS") || opt_get("MFM_SEQ_ROWS")) return MFM_ERR_UNSUPPORTED;       // tuning overrides keep the plain launches
  const int want[4] = {8, 2, 20, 30};
  if (L.count != 4 || !LD.row_path || LD.nch != 4 || LD.pre || LD.B != L.B) return MFM_ERR_UNSUPPORTED;
  for (int i = 0; i < 4; ++i)
    if (L.d[i].hk4 != want[i] || L.d[i].is_dec) return MFM_ERR_UNSUPPORTED;
  if (!((long)L.count * L.B < 6L * device_cus())) return MFM_ERR_UNSUPPORTED;          // one-row tiles only
  int max_threads = 64, total = 0;
  for (int i = 0; i < 4; ++i) {
    if (8 * L.d[i].Hp > max_threads) max_threads = 8 * L.d[i].Hp;
    L.d[i].block_begin = total; total += L.B;
  }
  if (max_threads < 1024) max_threads = 1024;       // the chain's work items are tabulated for up to 1024 threads
  L.n_img_blocks = 0;
  const bool img_full = !(opt_get("MFM_WT_IMG_FULL") && atoi(opt_get("MFM_WT_IMG_FULL")) == 0);
  if (!bwd && L.n_img > 0 && (device_cus() - total >= 8 || img_full)) {
    L.img_begin = total; L.n_img_blocks = device_cus() - total >= 8 ? std::min(device_cus() - total, 64) : 32; total += L.n_img_blocks;
  }
  else L.n_img = 0;
  size_t lds_bytes = small_lds_bytes(L, bwd, 1);
  const size_t lat = ((size_t)MFM_LAT_MAXSTAGES * MFM_LAT_ROW_THREADS * 4 + (bwd ? 2 : 1) * (size_t)LD.rec_size) * sizeof(float);
  if (lat > lds_bytes) lds_bytes = lat;
  if (lds_bytes > 160 * 1024) return MFM_ERR_UNSUPPORTED;
  if (bwd) {
    MFM_HIP_CHECK(hipFuncSetAttribute((const void*)lstm_seq_small_fold_kernel<true, 8, 2, 20, 30>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    MFM_LAUNCH_TIMED((lstm_seq_small_fold_kernel<true, 8, 2, 20, 30>), dim3(total), dim3(max_threads), lds_bytes, stream, L, LD, params, grads);
  } else {
    MFM_HIP_CHECK(hipFuncSetAttribute((const void*)lstm_seq_small_fold_kernel<false, 8, 2, 20, 30>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    MFM_LAUNCH_TIMED((lstm_seq_small_fold_kernel<false, 8, 2, 20, 30>), dim3(total), dim3(max_threads), lds_bytes, stream, L, LD, params, grads);
  }
  MFM_LAUNCH_CHECK("lstm_seq_small_fold_kernel");
  return MFM_OK;
}

}  // namespace mfm

