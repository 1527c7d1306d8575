// Whole-sequence LSTM recurrence with bf16 MFMA operands (BASELINE.json configs 2-4: "bf16 compute, fp32
// master weights / cell state / loss"), forward and BPTT, for gfx950.
//
// Same decomposition as lstm_seq.hip -- one persistent workgroup per (LSTM, 16-row batch tile), wave w owns
// hidden units [16w, 16w+16) of all four gates, weights resident in VGPRs for all T steps, the accumulator
// lane holds gates i,f,g,o of 4 units for one batch row so the pointwise math is lane-local and c stays in
// fp32 registers -- but the recurrent product runs on v_mfma_f32_16x16x32_bf16:
//   * operands: the weights are rounded to bf16 ONCE per launch (from the fp32 master copy; the decoder's
//     W_ih + W_hh is summed in fp32 and rounded once), h_{t-1} / dA_t are rounded when they are written to
//     the LDS exchange panel; products are exact, accumulation is fp32, everything saved for the backward
//     (activated gates, h, c) stays fp32 in HBM;
//   * 8 bf16 per lane and k-block instead of one fp32: h = 120 needs 16 MFMAs of 16 (bf16) cycles per step
//     where the fp32 kernel issues 120 of 32 cycles -- the matrix work drops from ~1.6 us to ~0.1 us per
//     step and the resident weights from 120 to 64 VGPRs;
//   * the exchange panel is [16 rows][K] bf16 with K contiguous, so a lane's MFMA B fragment (8 consecutive
//     k of one batch row) is ONE ds_read_b128 and its four new h values are ONE ds_write_b64; rows are
//     padded by 16 bytes, which spreads the 16 rows of a fragment read over all 64 banks;
//   * round 5: the step's global traffic is COALESCED through LDS staging (a tile's 16 rows are contiguous in every
//     slab): see the note in seqb_fwd_body; the waves a launch has beyond an LSTM's Hp / 16 leave after the prologue;
//     the time loops are branch-free and entered after a peeled step so that every vmcnt wait is a counted one.
// The k index of a fragment element is (k-block, lane >> 4, j); A and B fragments are built with the same
// convention, and the MFMA sums over all of k, so the result does not depend on the hardware's internal
// k order.
#include <algorithm>
#include <type_traits>

#include <stdlib.h>

#include "internal.h"
#include "lstm_seq_dev.h"
#include "pack_dev.h"

namespace mfm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// Debug build only (-DMFM_SEQB_STAMP=k, scripts/seqb_step_timeline.sh): a clock on the links of the forward step -- every wave
// adds up, over the steps t >= 1, the shader-clock distance from the top of the step to point k (ISSUE time: s_memtime does
// not wait for the vector pipes) and leaves the average in the last step's cell-state record (row 16 tile, unit = wave).
// Points: 1 LDS reads returned, 2 first barrier passed, 3 global loads + stores issued, 4 recurrent product issued, 5 gates /
// c / h computed, 6 LDS writes done (incl. the wait for the prefetched x-projection), 7 second barrier passed.
#ifndef MFM_SEQB_STAMP
#define MFM_SEQB_STAMP 0
#endif
#if MFM_SEQB_STAMP
#define SEQB_PT(k, ...) do { if (MFM_SEQB_STAMP == (k)) { __VA_ARGS__; stamp1 = __builtin_readcyclecounter(); } } while (0)
#else
#define SEQB_PT(k, ...) do { } while (0)
#endif

__device__ __forceinline__ f32x4 ld4b(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4b(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

__device__ __forceinline__ bf16x8 pack8(const float (&v)[8]) {
  const f32x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
  const bf16x4 a = __builtin_convertvector(lo, bf16x4), b = __builtin_convertvector(hi, bf16x4);
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ f32x4 mma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// Global traffic of the time loops goes through BUFFER instructions on one-time-step slabs: a piece that has nothing
// to load or store (batch row >= B, the first / last step's absent neighbour) falls beyond the end of the slab (or the
// slab is given 0 bytes), which the hardware range check turns into "load 0" / "drop the store".  So no memory
// instruction of a step sits under a branch, every step issues the same list of them, and the compiler can count: the
// wait for the NEXT step's prefetched pieces is vmcnt(#younger stores) instead of vmcnt(0).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t slab(const float* base, int64_t elem_off, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)(base + elem_off), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 bld4(__amdgpu_buffer_rsrc_t r, int off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ void bst4(__amdgpu_buffer_rsrc_t r, int off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 0);
}
// bf16-resident buffers (SeqDev::store_bf16): the same four values travel as 8 bytes; ST selects the element type, `off`
// is a BYTE offset in both forms
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <bool ST>
__device__ __forceinline__ f32x4 bldv(__amdgpu_buffer_rsrc_t r, int off) {
  if constexpr (ST) {
    const bf16x4 v = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
    return __builtin_convertvector(v, f32x4);
  } else {
    return bld4(r, off);
  }
}
// four stored values as raw registers (8 bytes bf16-resident, 16 bytes fp32) and their conversion
template <bool ST> struct RawV { typedef f32x4 type; };
template <> struct RawV<true> { typedef u32x2 type; };
template <bool ST>
__device__ __forceinline__ f32x4 cvt_raw(typename RawV<ST>::type v) {
  if constexpr (ST) return __builtin_convertvector(__builtin_bit_cast(bf16x4, v), f32x4);
  else return v;
}
template <bool ST>
__device__ __forceinline__ void bstv(__amdgpu_buffer_rsrc_t r, int off, f32x4 v) {
  if constexpr (ST) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, __builtin_convertvector(v, bf16x4)), r, off, 0, 0);
  else bst4(r, off, v);
}
template <bool ST>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t slabv(const float* base, int64_t elem_off, int bytes) {
  // elem_off counts ELEMENTS of the buffer's own type (bf16 when ST)
  return __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(base) + elem_off * (ST ? 2 : 4)), 0, bytes, 0x00020000);
}

// --------------------------------------------------------------------------------- forward
// KB = number of 32-wide k-blocks covering the hidden size (h <= 32 KB).
template <int KB, int KIND, bool ST>
__device__ __forceinline__ void seqb_fwd_body(const SeqDev& d, const int T, const int B, const int tile, __bf16* lds) {
  constexpr int ES = ST ? 2 : 4;          // bytes per stored gate / h element
  constexpr int HKP = KB * 32;
  constexpr int LROW = HKP + 8;      // bf16 elements per LDS row (16-byte pad)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int bi = lane & 15, q = lane >> 4;
  const int h = d.h, Hp = d.Hp;
  const bool active = wave < (Hp >> 4);
  const int u0 = wave * 16;
  const int b = tile * 16 + bi;
  const bool bvalid = active && (b < B);
  constexpr bool dec = KIND != 0;          // a launch holds LSTMs of one kind (seq_bf16_launch)

  bf16x8 w[4][KB];
  // MODE 0: W_hh (encoder)   1: W_ih (decoder step 0)   2: W_ih + W_hh (decoder steps >= 1; one rounding)
  auto load_w = [&](auto mode) {
    constexpr int MODE = decltype(mode)::value;
    const int unit = u0 + bi;
    const int uc = min(unit, h - 1);
    const int uok = (int)active & (int)(unit < h);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = kb * 32 + 8 * q + j;
          const int off = (g * h + uc) * h + min(k, h - 1);       // always a valid address
          const float m = (float)(uok & (int)(k < h));            // 0 for pad elements
          float x;
          if constexpr (MODE == 0) x = d.w_hh[off];
          else if constexpr (MODE == 1) x = d.w_ih[off];
          else x = d.w_ih[off] + d.w_hh[off];
          v[j] = x * m;
        }
        w[g][kb] = pack8(v);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // packed fragments (mfm_lstm_pack_bf16): lane-linear, one coalesced 16-byte load per fragment
  const bf16x8* pk = reinterpret_cast<const bf16x8*>(d.w_pack);
  const int64_t pack_frags = (int64_t)(Hp >> 4) * 4 * KB * 64;
  auto load_packed = [&](int which) {
    const bf16x8* base = pk + which * pack_frags + (int64_t)min(wave, (Hp >> 4) - 1) * 4 * KB * 64 + lane;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) w[g][kb] = base[(g * KB + kb) * 64];
  };
  if (pk) load_packed(dec ? 1 : 0);
  else if (dec) load_w(std::integral_constant<int, 1>{}); else load_w(std::integral_constant<int, 0>{});

  f32x4 bias[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    bias[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (dec && active) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int unit = u0 + 4 * q + r;
        if (unit < h) bias[g][r] = d.b_ih[g * h + unit] + d.b_hh[g * h + unit];
      }
    }
  }

  // exchange panel [2][16][LROW]: zero everywhere first (k beyond h meets zero weights, but 0 * garbage may be NaN)
  for (int idx = tid; idx < 2 * 16 * LROW; idx += blockDim.x) lds[idx] = (__bf16)0.0f;
  __syncthreads();
  if (dec) {
    for (int idx = tid; idx < 16 * h; idx += blockDim.x) {
      const int r = idx / h, unit = idx - r * h, br = tile * 16 + r;
      if (br < B) lds[r * LROW + unit] = (__bf16)d.h_init[(int64_t)br * d.ld_init + unit];
    }
    __syncthreads();
  }
  // The block is sized for the LARGEST LSTM of the launch; the waves beyond this LSTM's Hp / 16 leave here.  A wave
  // that has ended no longer counts at s_barrier, and the panel columns they would have covered (k in [Hp, HKP))
  // keep the zeros written above.  (They used to run the gate math on zeros: 47 % of the forward launch's waves at
  // the MOSI shapes -- VALU and issue slots taken from the waves that carry rows.)
  if (__builtin_amdgcn_readfirstlane((int)active) == 0) return;

  // pointers and slab geometry are formed once (a descriptor field read inside the loop is re-fetched from the
  // kernel-argument segment after every store: the compiler cannot prove the stores do not alias it)
  float* const gates_p = d.gates;
  float* const cs_p = d.cs;
  float* const hs_p = d.hs;
  const int64_t row4 = 4 * (int64_t)Hp;
  const int slab_g = B * 4 * Hp * ES, slab_h = B * Hp * 4;                   // bytes of one time step (gates; cs)
  const int slab_hs = B * Hp * ES;                                            // (hs)
  const int voff_h = bvalid ? (b * Hp + u0 + 4 * q) * 4 : slab_h;

  // ---- COALESCED global traffic (round 5).  A lane's share of a step's record is 8-16 bytes of one batch row per gate, so a
  // store instruction used to touch 16 cache lines with 32-64 bytes each: 768 line visits per workgroup and step (h = 120)
  // where the record has 224 lines, and the texture-address unit of the CU, which takes ~3 cycles per line, was busy for
  // ~3.8 k of the step's 4.4 k cycles (2.2 k of them the stores that every wave has to ISSUE before it reaches the barrier;
  // scripts/seqb_step_timeline.py, profiles/r05_seq_bf16_study.txt section 7).  But a tile's rows are CONTIGUOUS in every one of
  // the step's slabs ([B][4 Hp], [B][Hp]: 16 rows back to back), so the record is staged in LDS in row order and leaves as
  // 16-byte pieces in memory order -- piece i of the tile by thread i: every line visited once, 4 (bf16-resident) store
  // instructions per wave instead of 6.  The x-projection of the next step comes in the same way (2 loads instead of 4).
  // Step t: [LDS reads: x-projection(t), record(t-1), h(t-1)] barrier [loads x-projection(t+1); stores record(t-1); product;
  // gates; LDS writes: h(t), record(t), x-projection(t+1)] barrier.  Two barriers per step (the staging areas are single:
  // doubled they would cost the second workgroup per CU), one more than before, ~470 cycles.
  const int nthr = (Hp >> 4) * 64;                        // the waves that are left
  const int RBg = 4 * Hp * ES, RBc = Hp * 4, RBh = Hp * ES;      // bytes of a row in the gates / c / h slabs
  const int LBg = RBg + 32, LBc = RBc + 32, LBh = RBh + 32;      // and in LDS (padded: 16 rows on different banks)
  unsigned char* const sm = reinterpret_cast<unsigned char*>(lds) + 2 * 16 * LROW * 2;
  unsigned char* const outg = sm;
  unsigned char* const outc = outg + 16 * LBg;
  unsigned char* const outh = outc + 16 * LBc;
  unsigned char* const ing = outh + 16 * LBh;             // encoders only
  constexpr int NG = ES;                                  // 16-byte pieces of the gates tile per thread: 16 RBg / 16 / nthr = ES
  int pg_l[NG], pg_g[NG];                                 // LDS / slab byte offsets of this thread's pieces
#pragma unroll
  for (int j = 0; j < NG; ++j) {
    const int off = (tid + j * nthr) * 16;
    const int r = off / RBg;
    pg_l[j] = r * LBg + (off - r * RBg);
    pg_g[j] = tile * 16 * RBg + off;                      // (rows past the batch fall beyond the slab: dropped / zero)
  }
  const int pc_l = ((tid * 16) / RBc) * LBc + (tid * 16) % RBc, pc_g = tile * 16 * RBc + tid * 16;
  // h: 16 RBh bytes = nthr pieces of (ES == 2 ? 8 : 16) bytes
  constexpr int HPB = ST ? 8 : 16;
  const int ph_l = ((tid * HPB) / RBh) * LBh + (tid * HPB) % RBh, ph_g = tile * 16 * RBh + tid * HPB;
  // this lane's cells in the staging areas (row bi, units u0 + 4 q ..)
  const int cg_l = bi * LBg + (u0 + 4 * q) * ES, cc_l = bi * LBc + (u0 + 4 * q) * 4, ch_l = bi * LBh + (u0 + 4 * q) * ES;
  typedef typename RawV<ST>::type raw_t;                  // 4 stored values (8 or 16 bytes)

  if constexpr (KIND == 0) {      // x-projection of step 0 into the staging area
    const __amdgpu_buffer_rsrc_t r0 = slabv<ST>(gates_p, 0, slab_g);
#pragma unroll
    for (int j = 0; j < NG; ++j) *reinterpret_cast<f32x4*>(ing + pg_l[j]) = bld4(r0, pg_g[j]);
    lds_barrier();
  }

  float c[4] = {0.f, 0.f, 0.f, 0.f};
  f32x4 h_keep = f32x4{0.f, 0.f, 0.f, 0.f};
  int cur = 0;
  // `wo`: bytes of the slabs that receive step t-1's record (0 at the first step: the stores are issued and dropped, so that
  // every step has the same list of pending memory operations -- see the note on counted waits in the backward body)
#if MFM_SEQB_STAMP
  unsigned long long stamp0 = 0, stamp1 = 0, stamp_sum = 0;
#endif
  auto step = [&](const int t, auto rec) {
#if MFM_SEQB_STAMP
    stamp0 = __builtin_readcyclecounter();
    stamp1 = stamp0;
#endif
    // ---- LDS reads
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if constexpr (KIND == 0) acc[g] = cvt_raw<ST>(*reinterpret_cast<const raw_t*>(ing + cg_l + g * Hp * ES));
      else acc[g] = bias[g];
    }
    f32x4 og[NG], oc;
    raw_t oh;
#pragma unroll
    for (int j = 0; j < NG; ++j) og[j] = *reinterpret_cast<const f32x4*>(outg + pg_l[j]);
    oc = *reinterpret_cast<const f32x4*>(outc + pc_l);
    oh = *reinterpret_cast<const raw_t*>(outh + ph_l);
    bf16x8 hv[KB];
    if constexpr (decltype(rec)::value) {
      const __bf16* hb = lds + cur * (16 * LROW) + bi * LROW + 8 * q;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) hv[kb] = *reinterpret_cast<const bf16x8*>(hb + kb * 32);
    }
    SEQB_PT(1, asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"));
    lds_barrier();
    SEQB_PT(2, asm volatile("" ::: "memory"));
    // ---- global: next step's x-projection first (it is waited for at the end of this step; the stores stay younger)
    f32x4 gxn[NG];
    if constexpr (KIND == 0) {
      const __amdgpu_buffer_rsrc_t rn = slabv<ST>(gates_p, (int64_t)min(t + 1, T - 1) * B * row4, slab_g);
#pragma unroll
      for (int j = 0; j < NG; ++j) gxn[j] = bld4(rn, pg_g[j]);
    }
    {
      const int64_t tp = (int64_t)t - 1;        // (t = 0: empty ranges, whose base is never dereferenced; NOT max(t - 1, 0))
      const __amdgpu_buffer_rsrc_t rg = slabv<ST>(gates_p, tp * B * row4, t > 0 ? slab_g : 0);
      const __amdgpu_buffer_rsrc_t rc = slab(cs_p, tp * B * Hp, t > 0 ? slab_h : 0);
      const __amdgpu_buffer_rsrc_t rh = slabv<ST>(hs_p, tp * B * Hp, t > 0 ? slab_hs : 0);
#pragma unroll
      for (int j = 0; j < NG; ++j) bst4(rg, pg_g[j], og[j]);
      bst4(rc, pc_g, oc);
      if constexpr (ST) __builtin_amdgcn_raw_buffer_store_b64(oh, rh, ph_g, 0, 0);
      else bst4(rh, ph_g, oh);
    }
    SEQB_PT(3, asm volatile("" ::: "memory"));
    // ---- recurrent product, gates
    if constexpr (decltype(rec)::value) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = mma_bf16(w[g][kb], hv[kb], acc[g]);
    }
    SEQB_PT(4, asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])));
    f32x4 gi, gf, gg, go, cv, hn4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      gi[r] = act_sigmoid(acc[0][r]);
      gf[r] = act_sigmoid(acc[1][r]);
      gg[r] = act_tanh(acc[2][r]);
      go[r] = act_sigmoid(acc[3][r]);
      c[r] = gf[r] * c[r] + gi[r] * gg[r];
      cv[r] = c[r];
      hn4[r] = go[r] * act_tanh(c[r]);
    }
    h_keep = hn4;
    SEQB_PT(5, asm volatile("" : "+v"(hn4), "+v"(gi), "+v"(gf), "+v"(gg), "+v"(go)));
    // ---- LDS writes: h_t for the product, the record for the write-out, the next x-projection
    {
      const f32x4 hz = (b < B) ? hn4 : f32x4{0.f, 0.f, 0.f, 0.f};
      __bf16* hp = lds + (cur ^ 1) * (16 * LROW) + bi * LROW + u0 + 4 * q;
      *reinterpret_cast<bf16x4*>(hp) = __builtin_convertvector(hz, bf16x4);
    }
    auto put = [&](unsigned char* at, f32x4 v) {
      if constexpr (ST) *reinterpret_cast<bf16x4*>(at) = __builtin_convertvector(v, bf16x4);
      else *reinterpret_cast<f32x4*>(at) = v;
    };
    put(outg + cg_l, gi); put(outg + cg_l + Hp * ES, gf); put(outg + cg_l + 2 * Hp * ES, gg); put(outg + cg_l + 3 * Hp * ES, go);
    *reinterpret_cast<f32x4*>(outc + cc_l) = cv;
    put(outh + ch_l, hn4);
    if constexpr (KIND == 0) {
#pragma unroll
      for (int j = 0; j < NG; ++j) *reinterpret_cast<f32x4*>(ing + pg_l[j]) = gxn[j];
    }
    SEQB_PT(6, asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"));
    lds_barrier();
    SEQB_PT(7, asm volatile("" ::: "memory"));
#if MFM_SEQB_STAMP
    if (t >= 1) stamp_sum += stamp1 - stamp0;
#endif
    cur ^= 1;
  };
  // the weights are in registers before the loop: a load still pending on the entry edge is waited for inside the
  // loop on every step, with a count that also covers the previous step's stores
  auto touch_w = [&]() {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) asm volatile("" : "+v"(w[g][kb]));
  };
  if constexpr (KIND != 0) {
    step(0, std::true_type{});                               // W_ih x h_init
    if (T > 1) {                                             // W_ih (step 0) -> W_ih + W_hh (steps >= 1)
      if (pk) load_packed(0); else load_w(std::integral_constant<int, 2>{});
    }
  } else {
    step(0, std::false_type{});                              // h_{-1} = 0: no recurrent term
  }
  touch_w();
  for (int t = 1; t < T; ++t) step(t, std::true_type{});
#if MFM_SEQB_STAMP
  if (T > 1) *reinterpret_cast<float*>(outc + (wave >> 2) * 0 + 0 * LBc + wave * 4) = (float)stamp_sum / (float)(T - 1);   // row 0, unit = wave
  lds_barrier();
#endif
  {   // the last step's record
    const __amdgpu_buffer_rsrc_t rg = slabv<ST>(gates_p, (int64_t)(T - 1) * B * row4, slab_g);
    const __amdgpu_buffer_rsrc_t rc = slab(cs_p, (int64_t)(T - 1) * B * Hp, slab_h);
    const __amdgpu_buffer_rsrc_t rh = slabv<ST>(hs_p, (int64_t)(T - 1) * B * Hp, slab_hs);
#pragma unroll
    for (int j = 0; j < NG; ++j) bst4(rg, pg_g[j], *reinterpret_cast<const f32x4*>(outg + pg_l[j]));
    bst4(rc, pc_g, *reinterpret_cast<const f32x4*>(outc + pc_l));
    if constexpr (ST) __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const raw_t*>(outh + ph_l), rh, ph_g, 0, 0);
    else bst4(rh, ph_g, *reinterpret_cast<const f32x4*>(outh + ph_l));
  }
  if (d.h_last) {      // fp32 copy of h_{T-1} (the latent stack / the MFN heads read it; hs itself may be bf16)
    const __amdgpu_buffer_rsrc_t rl = slab(d.h_last, 0, slab_h);
    bst4(rl, voff_h, h_keep);
  }
}

// --------------------------------------------------------------------------------- backward
// The reduction runs over the gate columns in the padded numbering [4][HKP] (HKP = 32 KB): 4 KB k-blocks.
template <int KB, int KIND, bool ST>
__device__ __forceinline__ void seqb_bwd_body(const SeqDev& d, const int T, const int B, const int tile, __bf16* lds) {
  constexpr int ES = ST ? 2 : 4;
  constexpr int HKP = KB * 32;
  constexpr int NKB = 4 * KB;
  constexpr int LROW = 4 * HKP + 8;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int bi = lane & 15, q = lane >> 4;
  const int h = d.h, Hp = d.Hp;
  const bool active = wave < (Hp >> 4);
  const int u0 = wave * 16;
  const int b = tile * 16 + bi;
  const bool bvalid = active && (b < B);
  constexpr bool dec = KIND != 0;

  bf16x8 wT[NKB];
  auto load_wT = [&](auto mode) {
    constexpr int MODE = decltype(mode)::value;
    const int unit = u0 + bi;   // A row = output unit of dh
    const int uc = min(unit, h - 1);
    const int uok = (int)active & (int)(unit < h);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = kb * 32 + 8 * q + j;       // gate column in the [4][HKP] numbering
        const int g = k / HKP, up = k % HKP;
        const int off = (g * h + min(up, h - 1)) * h + uc;   // always a valid address
        const float m = (float)(uok & (int)(up < h));
        float x;
        if constexpr (MODE == 0) x = d.w_hh[off];
        else if constexpr (MODE == 1) x = d.w_ih[off];
        else x = d.w_ih[off] + d.w_hh[off];
        v[j] = x * m;
      }
      wT[kb] = pack8(v);
      if ((kb & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  };
  const bf16x8* pk = reinterpret_cast<const bf16x8*>(d.w_pack);
  const int64_t pack_frags = (int64_t)(Hp >> 4) * NKB * 64;          // == the forward packs' size
  const int bwd_first_pack = dec ? 2 : 1;                            // enc: [fwd, bwd]   dec: [fwd main, fwd first, bwd main, bwd first]
  auto load_packedT = [&](int which) {
    const bf16x8* base = pk + which * pack_frags + (int64_t)min(wave, (Hp >> 4) - 1) * NKB * 64 + lane;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) wT[kb] = base[kb * 64];
  };
  if (pk) load_packedT(bwd_first_pack);
  else if (dec) load_wT(std::integral_constant<int, 2>{}); else load_wT(std::integral_constant<int, 0>{});

  for (int idx = tid; idx < 2 * 16 * LROW; idx += blockDim.x) lds[idx] = (__bf16)0.0f;
  __syncthreads();
  if (__builtin_amdgcn_readfirstlane((int)active) == 0) return;       // idle waves leave (see the forward body)

  float* const gates_p = d.gates;
  const float* const cs_p = d.cs;
  const float* const dh_p = d.dh_ext;
  const float* const dc_p = d.dc_ext;
  const int64_t row4 = 4 * (int64_t)Hp;
  const int slab_g = B * 4 * Hp * ES, slab_h = B * Hp * 4;                   // bytes of one time step (gates; cs, dc_ext)
  const int slab_dh = B * Hp * ES;                                            // (the decoders' dh_ext)
  const int dhe_bytes = dec ? slab_dh : 0;          // per-step external dh exists for decoders only
  const bool has_dce = dc_p != nullptr;             // optional external dc (MFN encoder LSTMs)
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  // encoders receive an external gradient on h_{T-1} only: it seeds the recurrent term
  f32x4 dh_rec = zero4;
  if (!dec && bvalid) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int unit = u0 + 4 * q + r;
      if (unit < h) dh_rec[r] = dh_p[(int64_t)b * d.ld_dh + unit];
    }
  }

  // ---- COALESCED global traffic, as in the forward body: a tile's rows are contiguous in every slab, so the saved activations
  // of a step arrive as 16-byte pieces in memory order (piece i of the tile by thread i) into LDS staging areas and every lane
  // picks its cells from there; dA leaves the same way.  Per wave and step: ES + 3 loads and ES stores instead of 8 and 4, every
  // cache line visited once.  c_{t-1} is not fetched twice: the two cell-state areas take turns (this step's c_{t-1} is the
  // next step's c_t).  bf16-resident: the dA pieces are read straight from the exchange panel of the previous step (the panel
  // IS bf16 dA, and its two halves alternate).
  // Step t: [LDS reads: cells of step t, pieces of dA(t+1)] barrier [loads for step t-1; stores dA(t+1); gate gradients;
  // LDS writes: panel, staging of step t-1] barrier [recurrent product].
  const int nthr = (Hp >> 4) * 64;
  const int RBg = 4 * Hp * ES, RBc = Hp * 4, RBh = Hp * ES;
  const int LBg = RBg + 32, LBc = RBc + 32, LBh = RBh + 32;
  unsigned char* const sm = reinterpret_cast<unsigned char*>(lds) + 2 * 16 * LROW * 2;
  unsigned char* const ing = sm;
  unsigned char* const cbuf0 = ing + 16 * LBg;
  unsigned char* const cbuf1 = cbuf0 + 16 * LBc;
  unsigned char* const dheb = cbuf1 + 16 * LBc;
  unsigned char* const outf = dheb + 16 * LBh;                     // fp32-resident only: dA cells in slab layout
  unsigned char* const dceb = outf + (ST ? 0 : 16 * LBg);          // only when the launch has a dc_ext (seq_bf16_launch)
  constexpr int NG = ES;
  int pg_l[NG], pg_g[NG], po_l[NG];        // gates pieces: staging / slab offsets; where the dA piece is read from
#pragma unroll
  for (int j = 0; j < NG; ++j) {
    const int off = (tid + j * nthr) * 16;
    const int r = off / RBg, rem = off - r * RBg;
    pg_l[j] = r * LBg + rem;
    pg_g[j] = tile * 16 * RBg + off;
    if constexpr (ST) {                    // panel row [4][HKP] bf16 (+ pad): gate g, column c of the slab row
      const int g = rem / (Hp * 2), cb = rem - g * (Hp * 2);
      po_l[j] = r * (LROW * 2) + g * (HKP * 2) + cb;
    } else {
      po_l[j] = pg_l[j];
    }
  }
  const int pc_l = ((tid * 16) / RBc) * LBc + (tid * 16) % RBc, pc_g = tile * 16 * RBc + tid * 16;
  constexpr int HPB = ST ? 8 : 16;
  const int ph_l = ((tid * HPB) / RBh) * LBh + (tid * HPB) % RBh, ph_g = tile * 16 * RBh + tid * HPB;
  const int cg_l = bi * LBg + (u0 + 4 * q) * ES, cc_l = bi * LBc + (u0 + 4 * q) * 4, ch_l = bi * LBh + (u0 + 4 * q) * ES;
  typedef typename RawV<ST>::type raw_t;

  // the pieces of one step into the staging areas (sizes 0: the loads return zeros -- c_{-1}, absent operands)
  struct Pieces { f32x4 g[NG]; f32x4 c; raw_t dh; f32x4 dc; };
  auto request = [&](const int t, Pieces& P) {
    if (has_dce) P.dc = bld4(slab(dc_p, (int64_t)max(t, 0) * B * Hp, t >= 0 ? slab_h : 0), pc_g);     // (first: see seq_bf16_launch)
    const __amdgpu_buffer_rsrc_t rg = slabv<ST>(gates_p, (int64_t)t * B * row4, t >= 0 ? slab_g : 0);
#pragma unroll
    for (int j = 0; j < NG; ++j) P.g[j] = bld4(rg, pg_g[j]);
    P.c = bld4(slab(cs_p, ((int64_t)t - 1) * B * Hp, t >= 1 ? slab_h : 0), pc_g);                      // c_{t-1}
    const __amdgpu_buffer_rsrc_t rd = slabv<ST>(dh_p, (int64_t)t * B * Hp, t >= 0 ? dhe_bytes : 0);
    if constexpr (ST) P.dh = __builtin_amdgcn_raw_buffer_load_b64(rd, ph_g, 0, 0);
    else P.dh = bld4(rd, ph_g);
  };
  auto deposit = [&](const Pieces& P, unsigned char* cdst) {
#pragma unroll
    for (int j = 0; j < NG; ++j) *reinterpret_cast<f32x4*>(ing + pg_l[j]) = P.g[j];
    *reinterpret_cast<f32x4*>(cdst + pc_l) = P.c;
    *reinterpret_cast<raw_t*>(dheb + ph_l) = P.dh;
    if (has_dce) *reinterpret_cast<f32x4*>(dceb + pc_l) = P.dc;
  };
  {   // step T-1: its activations, c_{T-1} and c_{T-2}
    Pieces P;
    request(T - 1, P);
    const f32x4 ct = bld4(slab(cs_p, (int64_t)(T - 1) * B * Hp, slab_h), pc_g);
    deposit(P, cbuf1);
    *reinterpret_cast<f32x4*>(cbuf0 + pc_l) = ct;
    lds_barrier();
  }
  float dc[4] = {0.f, 0.f, 0.f, 0.f};
  int cur = 0, ci = 0;               // panel half of this step; cell-state area that holds c_t (the other one: c_{t-1})

  // REC: the step hands dA_t to the recurrent product (every step of a decoder, steps >= 1 of an encoder);
  // SWAP: decoder step 0, whose product runs through W_ih only
  auto step = [&](const int t, auto rec, auto swap) {
    unsigned char* const ca = ci ? cbuf1 : cbuf0;
    unsigned char* const cb = ci ? cbuf0 : cbuf1;
    // ---- LDS reads: this lane's cells of step t, this thread's pieces of dA(t+1)
    const f32x4 gi = cvt_raw<ST>(*reinterpret_cast<const raw_t*>(ing + cg_l));
    const f32x4 gf = cvt_raw<ST>(*reinterpret_cast<const raw_t*>(ing + cg_l + Hp * ES));
    const f32x4 gg = cvt_raw<ST>(*reinterpret_cast<const raw_t*>(ing + cg_l + 2 * Hp * ES));
    const f32x4 go = cvt_raw<ST>(*reinterpret_cast<const raw_t*>(ing + cg_l + 3 * Hp * ES));
    const f32x4 ct = *reinterpret_cast<const f32x4*>(ca + cc_l);
    const f32x4 cp = *reinterpret_cast<const f32x4*>(cb + cc_l);
    const f32x4 dh = dh_rec + cvt_raw<ST>(*reinterpret_cast<const raw_t*>(dheb + ch_l));
    f32x4 dce = zero4;
    if (has_dce) dce = *reinterpret_cast<const f32x4*>(dceb + cc_l);
    f32x4 og[NG];
    {
      const unsigned char* src = ST ? reinterpret_cast<const unsigned char*>(lds + (cur ^ 1) * (16 * LROW)) : outf;
#pragma unroll
      for (int j = 0; j < NG; ++j) og[j] = *reinterpret_cast<const f32x4*>(src + po_l[j]);
    }
    lds_barrier();
    // ---- global: the loads first (they are waited for before the second barrier; the stores stay younger)
    Pieces P;
    request(t - 1, P);
    {
      const __amdgpu_buffer_rsrc_t ro = slabv<ST>(gates_p, ((int64_t)t + 1) * B * row4, t + 1 < T ? slab_g : 0);
#pragma unroll
      for (int j = 0; j < NG; ++j) bst4(ro, pg_g[j], og[j]);
    }
    // ---- gate gradients
    f32x4 dai, daf, dag, dao;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float tc = act_tanh(ct[r]);
      const float dot = dh[r] * tc;
      const float dct = dh[r] * go[r] * (1.0f - tc * tc) + dc[r] + dce[r];
      dai[r] = dct * gg[r] * gi[r] * (1.0f - gi[r]);
      daf[r] = dct * cp[r] * gf[r] * (1.0f - gf[r]);
      dag[r] = dct * gi[r] * (1.0f - gg[r] * gg[r]);
      dao[r] = dot * go[r] * (1.0f - go[r]);
      dc[r] = dct * gf[r];
    }
    // ---- LDS writes: the panel (bf16 dA: the product's operand, and what a bf16-resident plan stores), the fp32 cells
    // otherwise, the staging of step t-1
    __bf16* db = lds + cur * (16 * LROW);
    {
      __bf16* dp = db + bi * LROW + u0 + 4 * q;
      *reinterpret_cast<bf16x4*>(dp) = __builtin_convertvector(dai, bf16x4);
      *reinterpret_cast<bf16x4*>(dp + HKP) = __builtin_convertvector(daf, bf16x4);
      *reinterpret_cast<bf16x4*>(dp + 2 * HKP) = __builtin_convertvector(dag, bf16x4);
      *reinterpret_cast<bf16x4*>(dp + 3 * HKP) = __builtin_convertvector(dao, bf16x4);
    }
    if constexpr (!ST) {
      *reinterpret_cast<f32x4*>(outf + cg_l) = dai; *reinterpret_cast<f32x4*>(outf + cg_l + Hp * ES) = daf;
      *reinterpret_cast<f32x4*>(outf + cg_l + 2 * Hp * ES) = dag; *reinterpret_cast<f32x4*>(outf + cg_l + 3 * Hp * ES) = dao;
    }
    deposit(P, ca);                        // c_{t-2} takes the place of c_t
    lds_barrier();
    if constexpr (decltype(rec)::value) {
      if constexpr (decltype(swap)::value) {                  // grad wrt the step-0 input goes through W_ih only
        if (pk) load_packedT(3); else load_wT(std::integral_constant<int, 1>{});
      }
      f32x4 a0 = zero4, a1 = zero4, a2 = zero4, a3 = zero4;
      const __bf16* dp = db + bi * LROW + 8 * q;
#pragma unroll
      for (int kb = 0; kb < NKB; kb += 4) {
        bf16x8 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const bf16x8*>(dp + (kb + j) * 32);
        a0 = mma_bf16(wT[kb + 0], v[0], a0);
        a1 = mma_bf16(wT[kb + 1], v[1], a1);
        a2 = mma_bf16(wT[kb + 2], v[2], a2);
        a3 = mma_bf16(wT[kb + 3], v[3], a3);
      }
      dh_rec = (a0 + a1) + (a2 + a3);
    }
    cur ^= 1;
    ci ^= 1;
  };
  // Branch-free loop entered after a peeled step with the weights already in registers: the pending-memory state on
  // the entry edge is the back edge's, see the forward body.
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) asm volatile("" : "+v"(wT[kb]));
  constexpr std::true_type yes{};
  constexpr std::false_type no{};
  if (T >= 2) {
    step(T - 1, yes, no);
    for (int t = T - 2; t >= 1; --t) step(t, yes, no);
  }
  if constexpr (KIND != 0) step(0, yes, yes); else step(0, no, no);
  {   // dA of step 0
    const unsigned char* src = ST ? reinterpret_cast<const unsigned char*>(lds + (cur ^ 1) * (16 * LROW)) : outf;
    const __amdgpu_buffer_rsrc_t ro = slabv<ST>(gates_p, 0, slab_g);
#pragma unroll
    for (int j = 0; j < NG; ++j) bst4(ro, pg_g[j], *reinterpret_cast<const f32x4*>(src + po_l[j]));
  }
  if (dec && bvalid && d.d_h_init) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int unit = u0 + 4 * q + r;
      if (unit < h) d.d_h_init[(int64_t)b * d.ld_dinit + unit] = dh_rec[r];
    }
  }
}

#define MFM_SEQB_CASES(BODY, KIND)                                        \
  switch (kb) {                                                           \
    case 1: BODY<1, KIND, ST>(d, L.T, L.B, tile, lds); break;             \
    case 2: BODY<2, KIND, ST>(d, L.T, L.B, tile, lds); break;             \
    case 3: BODY<3, KIND, ST>(d, L.T, L.B, tile, lds); break;             \
    case 4: BODY<4, KIND, ST>(d, L.T, L.B, tile, lds); break;             \
    default: break;                                                       \
  }

// ST: bf16-resident saved activations (every LSTM of a launch has the same SeqDev::store_bf16)
template <bool BWD, int KIND, bool ST>
__global__ __launch_bounds__(512) void lstm_seq_bf16_kernel(const SeqLaunch L) {
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  int di = 0;
  const int bid = blockIdx.x;
#pragma unroll 1
  for (int i = 1; i < L.count; ++i)
    if (bid >= L.d[i].block_begin) di = i;
  const SeqDev& d = L.d[di];
  const int tile = bid - d.block_begin;
  const int kb = (d.h + 31) >> 5;
  if (BWD) { MFM_SEQB_CASES(seqb_bwd_body, KIND) } else { MFM_SEQB_CASES(seqb_fwd_body, KIND) }
}

// --------------------------------------------------------------------------------- weight packing
// The fragments above are gathers over the fp32 weight matrices (8 consecutive k of one gate row forward, 8 gate
// rows of one column backward): fetched inside the recurrence kernels they cost 128-256 uncoalesced load
// instructions per lane and dominate a 20-step launch (measured: 73-110 us per launch at B=32, ~5 us of it steps).
// So the bf16 fragments are built ONCE per step by a fully parallel kernel into a lane-linear image -- fragment f of
// wave w is one 16-byte element at [(w * nfrag + f) * 64 + lane] -- and the recurrences start with 16-32 coalesced
// loads.  Pack order per LSTM: encoder [fwd W_hh | bwd W_hh]; decoder [fwd W_ih+W_hh | fwd W_ih | bwd W_ih+W_hh |
// bwd W_ih].  Every pack has (Hp/16) * 4 * KB * 64 fragments.
// (the structs and the device body live in pack_dev.h: the step's three weight images are packed by ONE launch)
__global__ __launch_bounds__(256) void lstm_pack_bf16_kernel(const PackLaunch L) {
  lstm_pack_body(L, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

// fills `out` for up to MFM_MAX_SEQ * 2 LSTMs (out->total = fragments, 0: nothing to pack)
int lstm_pack_prepare(const MfmSeqDesc* descs, int count, PackLaunch* out) {
  memset(out, 0, sizeof(*out));
  MFM_REQUIRE(descs && count >= 0 && count <= MFM_MAX_SEQ * 2, "lstm pack: %d descriptors", count);
  int64_t total = 0;
  for (int i = 0; i < count; ++i) {
    const MfmSeqDesc& s = descs[i];
    if (s.h > MFM_SEQ_MAX_RESIDENT_H) continue;      // step-by-step fp32 path: nothing to pack
    MFM_REQUIRE(s.h >= 1 && s.w_hh && s.w_pack, "mfm_lstm_pack_bf16: h=%d, w_hh / w_pack must be set", s.h);
    if (s.is_dec) MFM_REQUIRE(s.w_ih, "mfm_lstm_pack_bf16: decoder needs w_ih");
    PackItem& it = out->it[out->count++];
    it.w_hh = s.w_hh; it.w_ih = s.w_ih; it.out = reinterpret_cast<pk_bf16x8*>(s.w_pack);
    it.h = s.h; it.Hp = round_up(s.h, 16); it.KB = cdiv(s.h, 32); it.is_dec = s.is_dec;
    MFM_REQUIRE(total < (int64_t)1 << 30, "mfm_lstm_pack_bf16: too many fragments");
    it.frag_begin = (int)total;
    total += (int64_t)(s.is_dec ? 4 : 2) * (it.Hp >> 4) * 4 * it.KB * 64;
  }
  out->total = total;
  return MFM_OK;
}

}  // namespace mfm

extern "C" int64_t mfm_lstm_pack_bytes(int32_t h, int32_t is_dec) {
  if (h < 1 || h > mfm::MFM_SEQ_MAX_RESIDENT_H) return 0;
  const int64_t Hp = (h + 15) / 16 * 16, KB = (h + 31) / 32;
  return (is_dec ? 4 : 2) * (Hp / 16) * 4 * KB * 64 * 16;
}

extern "C" int mfm_lstm_pack_bf16(const MfmSeqDesc* descs, int count, void* stream) {
  using namespace mfm;
  MFM_REQUIRE(descs && count >= 1, "mfm_lstm_pack_bf16: no descriptors");
  for (int done = 0; done < count; done += MFM_MAX_SEQ * 2) {
    PackLaunch L;
    const int rc = lstm_pack_prepare(descs + done, std::min(count - done, MFM_MAX_SEQ * 2), &L);
    if (rc != MFM_OK) return rc;
    if (L.count == 0) continue;
    MFM_LAUNCH_TIMED(lstm_pack_bf16_kernel, dim3((unsigned)((L.total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, L);
    MFM_LAUNCH_CHECK("lstm_pack_bf16_kernel");
  }
  return MFM_OK;
}

namespace mfm {

// L: descriptors filled by seq_launch (lstm_seq.hip); every LSTM has h <= MFM_SEQ_MAX_RESIDENT_H.
int seq_bf16_launch(SeqLaunch& L, bool bwd, hipStream_t stream) {
  const int tiles = cdiv(L.B, 16);
  for (int kind = 0; kind < 2; ++kind) {
    SeqLaunch K = L;
    K.count = 0;
    int ktotal = 0, max_waves = 1;
    size_t lds_bytes = 0;
    for (int i = 0; i < L.count; ++i) {
      if ((L.d[i].is_dec != 0) != (kind == 1)) continue;
      K.d[K.count] = L.d[i];
      K.d[K.count].block_begin = ktotal;
      ktotal += tiles;
      ++K.count;
      const SeqDev& d = L.d[i];
      if (d.Hp / 16 > max_waves) max_waves = d.Hp / 16;
      const size_t hkp = (size_t)((d.h + 31) / 32) * 32;
      size_t need = 2 * 16 * ((bwd ? 4 * hkp : hkp) + 8) * sizeof(__bf16);
      {   // staging areas of the coalesced traffic (seqb_fwd_body / seqb_bwd_body)
        const size_t es = d.store_bf16 ? 2 : 4;
        const size_t lbg = 4 * d.Hp * es + 32, lbc = d.Hp * 4 + 32, lbh = d.Hp * es + 32;
        if (!bwd) need += 16 * (lbg + lbc + lbh) + (kind == 0 ? 16 * lbg : 0);
        else need += 16 * (lbg + 2 * lbc + lbh) + (d.store_bf16 ? 0 : 16 * lbg) + (d.dc_ext ? 16 * lbc : 0);
      }
      if (need > lds_bytes) lds_bytes = need;
    }
    if (K.count == 0) continue;
    const dim3 grid(ktotal), block(64 * max_waves);
    const bool st = K.d[0].store_bf16 != 0;
    for (int i = 1; i < K.count; ++i)
      MFM_REQUIRE((K.d[i].store_bf16 != 0) == st, "lstm_seq (bf16): the LSTMs of one launch must agree on store_bf16");
#define MFM_SEQB_GO(BWD_, KIND_)                                                                                          \
  do {                                                                                                                    \
    static bool big_lds = false;                 /* (once per instantiation and process: not a stream operation) */       \
    if (lds_bytes > 64 * 1024 && !big_lds) {                                                                              \
      MFM_HIP_CHECK(hipFuncSetAttribute((const void*)lstm_seq_bf16_kernel<BWD_, KIND_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));  \
      MFM_HIP_CHECK(hipFuncSetAttribute((const void*)lstm_seq_bf16_kernel<BWD_, KIND_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      big_lds = true;                                                                                                     \
    }                                                                                                                     \
    MFM_REQUIRE(lds_bytes <= 160 * 1024, "lstm_seq (bf16): %zu bytes of LDS", lds_bytes);                                 \
    if (st) MFM_LAUNCH_TIMED((lstm_seq_bf16_kernel<BWD_, KIND_, true>), grid, block, lds_bytes, stream, K);             \
    else MFM_LAUNCH_TIMED((lstm_seq_bf16_kernel<BWD_, KIND_, false>), grid, block, lds_bytes, stream, K);               \
  } while (0)
    if (bwd) {
      if (kind) MFM_SEQB_GO(true, 1); else MFM_SEQB_GO(true, 0);
    } else {
      if (kind) MFM_SEQB_GO(false, 1); else MFM_SEQB_GO(false, 0);
    }
#undef MFM_SEQB_GO
    MFM_LAUNCH_CHECK(bwd ? "lstm_seq_bf16_bwd_kernel" : "lstm_seq_bf16_fwd_kernel");
  }
  return MFM_OK;
}

}  // namespace mfm
