// Weight-gradient products at SMALL row counts (round 2): C[m, n] += alpha * sum_r A[r, m] B[r, n], both operands
// row-major over the rows r = (t, b) that are summed away -- every product of the step's last launch has this shape
// (LSTM dW_ih / dW_hh / bias column sums, decoder fc1, the latent stack's Linears, the MFN's).
//
// The grouped GEMM (gemm.hip) walks the rows 32 at a time through a register ring: at T*B = 640 a 32x32 tile is a chain
// of 20 dependent load round trips, and the launch takes 24.7 us (MFM_KL_EF) / 41.7 us (MFM_KL) for a few MFLOP.
// Here the rows are cut into chunks of <= 160 and a workgroup owns (tile, chunk): it requests its whole 160 x 32 slices
// of A and of B at once (10 16-byte loads per thread), writes them to LDS, runs 40 MFMA steps per wave (each wave one
// 16x16 fragment of the 32x32 tile, no cross-wave reduction) and adds its partial tile with atomics, which the
// accumulating problems of the grouped GEMM did anyway.  One memory round trip per workgroup instead of twenty.
//
// LDS images are [row][32] with odd rows rotated by 16 columns: the two 32-lane halves of an operand-fragment read
// (rows 4 ks + {0, 1} / {2, 3}, 16 columns each) then touch 32 distinct banks, and the images need no padding
// (40 KB per workgroup: three per CU).
#include <stdlib.h>

#include <algorithm>

#include "internal.h"
#include "gemm_common.h"

namespace mfm {

namespace {

constexpr int TN_T = 32;          // tile edge

// The kernel's own problem table (round 4): only what a TN product needs, so that MFM_TN_MAXP = 112 problems fit one kernel
// argument block (12 KB; the grouped GEMM's table is 9.6 KB for 56): the ~90 weight-gradient products of the MFN plans
// (six LSTMs, three decoders + fc1, eight attention Linears, 22 latent Linears) leave in ONE launch instead of two.
struct TnProblem {
  const float* a; const float* b; float* c; float* c2; float* csum;
  int a_sz, b_sz, c_sz, a_sk, b_sk, ldc, m, n_valid, k, batch, tiles_m, tiles_n, block_begin, k_per_split, a_bf16;
  float alpha;
};
struct TnGroup { TnProblem p[MFM_TN_MAXP]; int begins[MFM_TN_MAXP]; int count; };

// TN_KC rows per chunk (80 / 160 / 320): 16-byte loads per thread and operand = TN_KC / 32
template <int TN_KC>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const TnGroup g) {
  constexpr int TN_LOADS = TN_KC * (TN_T / 4) / 256;
  extern __shared__ __attribute__((aligned(16))) float tn_lds[];
  float* As = tn_lds;
  float* Bs = tn_lds + TN_KC * TN_T;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int bi = lane & 15, q = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;

  int pi = 0;
  const int bid = blockIdx.x;
#pragma unroll
  for (int i = 1; i < MFM_TN_MAXP; ++i) pi += (bid >= g.begins[i]) ? 1 : 0;
  const TnProblem& P = g.p[pi];
  const TnProblem& d = P;
  int local = bid - P.block_begin;
  {
    // workgroups go to the 8 XCDs round-robin by id and every XCD has its own L2: give each XCD one contiguous run of a
    // problem's logical (chunk, tile) order, as gemm.hip does, so that the tiles sharing an operand slice meet in one L2
    // (without it the launch pulled 45 MB through the fabric for ~8 MB of operands)
    constexpr int NX = 8;
    const int nb = ((pi + 1 < g.count) ? g.begins[pi + 1] : (int)gridDim.x) - P.block_begin;
    const int x = local % NX, j = local / NX;
    const int per = nb / NX, rem = nb % NX;
    local = x * per + (x < rem ? x : rem) + j;
  }
  const int tn = local % P.tiles_n; local /= P.tiles_n;
  const int tm = local % P.tiles_m; local /= P.tiles_m;
  const int z = local % d.batch;
  const int split = local / d.batch;
  const int m0 = tm * TN_T, n0 = tn * TN_T;
  const int kbeg = split * P.k_per_split;
  const int klen = min(d.k - kbeg, P.k_per_split);          // 1 .. TN_KC rows of this chunk

  const float* A = d.a + (int64_t)z * d.a_sz;
  const float* __restrict__ Bm = d.b + (int64_t)z * d.b_sz;
  const int a_sk = (int)d.a_sk, b_sk = (int)d.b_sk;
  // descriptors end at the last valid element: a 16-byte group that runs past it (or past its row's valid columns)
  // returns zeros / neighbours' values for the excess elements, which only reach outputs that are never stored
  // a_bf16 (bf16-resident plans: the decoders' dA buffers): A holds __bf16 elements, addressed by the same element strides
  const bool a16 = d.a_bf16 != 0;
  if (a16) A = reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(d.a) + (int64_t)z * d.a_sz);
  const int a_elems = (d.m - 1) + (d.k - 1) * a_sk + 1;
  const int a_bytes = a16 ? ((a_elems * 2 + 3) & ~3) : a_elems * 4;
  const int b_bytes = ((max(d.n_valid, 1) - 1) + (d.k - 1) * b_sk + 1) * 4;
  const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t bres = __builtin_amdgcn_make_buffer_rsrc((void*)Bm, 0, b_bytes, 0x00020000);

  // ---- all operand bytes of the workgroup are requested before anything waits
  f32x4 ra[TN_LOADS], rb[TN_LOADS];
#pragma unroll
  for (int j = 0; j < TN_LOADS; ++j) {
    const int idx = tid + j * 256;
    const int r = idx >> 3, c4 = idx & 7;
    const bool rok = r < klen;
    const int offa = (rok & (m0 + 4 * c4 < d.m)) ? ((kbeg + r) * a_sk + m0 + 4 * c4) * 4 : -16;
    const int offb = (rok & (n0 + 4 * c4 < d.n_valid)) ? ((kbeg + r) * b_sk + n0 + 4 * c4) * 4 : -16;
    if (a16) {        // (wave-uniform) four bf16 values = one 8-byte load
      typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
      typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
      const u32x2_t raw = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(ares, offa >= 0 ? offa >> 1 : -16, 0, 0));
      ra[j] = __builtin_convertvector(__builtin_bit_cast(bf16x4_t, raw), f32x4);
    } else {
      ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ares, offa, 0, 0));
    }
    rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bres, offb, 0, 0));
  }
  // elements past the valid columns of a straddling group hold the neighbours' values: zero them, so that no Inf / NaN
  // of an unrelated buffer can meet a zero of the other operand
#pragma unroll
  for (int j = 0; j < TN_LOADS; ++j) {
    const int idx = tid + j * 256;
    const int r = idx >> 3, c4 = idx & 7;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ra[j][e] = (m0 + 4 * c4 + e < d.m) ? ra[j][e] : 0.0f;
      rb[j][e] = (n0 + 4 * c4 + e < d.n_valid) ? rb[j][e] : 0.0f;
    }
    const int sw = (4 * c4 + 16 * (r & 1)) & 31;
    *reinterpret_cast<f32x4*>(As + r * TN_T + sw) = ra[j];
    *reinterpret_cast<f32x4*>(Bs + r * TN_T + sw) = rb[j];
  }
  __syncthreads();

  // ---- optional: column sums of the A slice (bias gradients riding on their weight gradient's product; internal.h).  The
  // condition is uniform over the workgroup.
  {
    float* csum = d.csum;
    if (csum && tn == 0 && z == 0) {
      __shared__ float cs[4][TN_T];
      const int c = tid & 31, part = tid >> 5;
      float sacc = 0.0f;
      for (int r = part; r < klen; r += 8) sacc += As[r * TN_T + ((c + 16 * (r & 1)) & 31)];
      sacc += __shfl_xor(sacc, 32, 64);              // the two row parts of a wave
      if (lane < 32) cs[wave][c] = sacc;
      __syncthreads();
      if (tid < TN_T && m0 + tid < d.m) atomicAdd(csum + m0 + tid, d.alpha * (cs[0][tid] + cs[1][tid] + cs[2][tid] + cs[3][tid]));
    }
  }

  // ---- wave (wm, wn): fragment rows m0 + 16 wm .., columns n0 + 16 wn ..
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  {
    const int rot = 16 * (q & 1);
    const float* ap = As + q * TN_T + ((16 * wm + bi + rot) & 31);
    const float* bp = Bs + q * TN_T + ((16 * wn + bi + rot) & 31);
    const int nks = (klen + 3) >> 2;               // rows klen .. 4 nks - 1 of the images are zeros (loaded out of range)
    int ks = 0;
    for (; ks + 4 <= nks; ks += 4) {
      float a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { a[u] = ap[(ks + u) * 4 * TN_T]; b[u] = bp[(ks + u) * 4 * TN_T]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = mma16x16x4(a[u], b[u], acc);
    }
    for (; ks < nks; ++ks) acc = mma16x16x4(ap[ks * 4 * TN_T], bp[ks * 4 * TN_T], acc);
  }

  // ---- partial tile -> C (and C2) with atomics
  float* __restrict__ C = d.c + (int64_t)z * d.c_sz;
  float* __restrict__ C2 = d.c2 ? d.c2 + (int64_t)z * d.c_sz : nullptr;
  const int col = n0 + 16 * wn + bi;
  if (col < d.n_valid) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + 16 * wm + 4 * q + r;
      if (row < d.m) {
        const float v = d.alpha * acc[r];
        atomicAdd(C + (int64_t)row * d.ldc + col, v);
        if (C2) atomicAdd(C2 + (int64_t)row * d.ldc + col, v);
      }
    }
  }
}

}  // namespace

// true when every problem is an accumulating (or, with c_is_zero, plain) TN product the kernel takes and the row counts are small enough for the
// chunking to pay (max_rows: the caller's crossover)
bool gemm_tn_supported(const MfmGemmDesc* descs, int count, int max_rows, bool c_is_zero) {
  if (count < 1 || count > MFM_TN_MAXP) return false;
  for (int i = 0; i < count; ++i) {
    const MfmGemmDesc& d = descs[i];
    const bool dbg = opt_get("MFM_PLAN_DEBUG") != nullptr;
    auto no = [&](const char* why) {
      if (dbg) fprintf(stderr, "[mfm gemm_tn] problem %d of %d declined: %s (m %d n %d k %d batch %d a_sm %lld a_sk %lld b_sk %lld b_sn %lld acc %d)\n", i, count, why,
                       d.m, d.n, d.k, d.batch, (long long)d.a_sm, (long long)d.a_sk, (long long)d.b_sk, (long long)d.b_sn, d.accumulate);
      return false;
    };
    if (!d.a || !d.b || !d.c || d.m < 1 || d.n < 1 || d.k < 1 || d.batch < 1) return no("empty");
    // a non-accumulating product (C = ...) is taken when the caller vouches that C holds zeros: 0 + v is v exactly
    if (d.a_sm != 1 || d.b_sn != 1 || (!d.accumulate && !c_is_zero) || d.bias || d.bias2) return no("not an accumulating TN product");
    if (d.k > max_rows) return no("too many rows");
    if (d.a_bf16 && ((d.a_sk & 3) || (d.a_sz & 3) || (reinterpret_cast<uintptr_t>(d.a) & 7))) return no("bf16 A not 8-byte shaped");
    if (d.c_bf16) return no("bf16 output");
    const int64_t lim = (int64_t)1 << 29;
    if ((int64_t)(d.k - 1) * d.a_sk + d.m >= lim || (int64_t)(d.k - 1) * d.b_sk + d.n >= lim) return false;
    if (d.a_sz >= lim || d.b_sz >= lim || d.c_sz >= lim || d.ldc >= lim) return no("strides beyond 2^29");
  }
  return true;
}

int gemm_tn_launch(const MfmGemmDesc* descs, int count, int max_rows, bool c_is_zero, hipStream_t stream) {
  MFM_REQUIRE(gemm_tn_supported(descs, count, max_rows, c_is_zero), "gemm tn: unsupported group (count %d)", count);
  int KC = 160;                                   // rows per chunk; MFM_GEMM_TN_KC=80|160|320 (tuning)
  if (const char* e = opt_get("MFM_GEMM_TN_KC")) { const int v = atoi(e); if (v == 80 || v == 160 || v == 320) KC = v; }
  TnGroup g;
  memset(&g, 0, sizeof(g));
  g.count = count;
  int total = 0;
  for (int i = 0; i < count; ++i) {
    const MfmGemmDesc& s = descs[i];
    TnProblem& P = g.p[i];
    P.a = s.a; P.b = s.b; P.c = s.c; P.c2 = s.c2; P.csum = gemm_get_colsum_host(s);
    P.a_sz = (int)s.a_sz; P.b_sz = (int)s.b_sz; P.c_sz = (int)s.c_sz; P.a_sk = (int)s.a_sk; P.b_sk = (int)s.b_sk; P.ldc = (int)s.ldc;
    P.m = s.m; P.k = s.k; P.batch = s.batch; P.a_bf16 = s.a_bf16; P.alpha = s.alpha;
    P.n_valid = (s.n_valid <= 0 || s.n_valid > s.n) ? s.n : s.n_valid;
    P.tiles_m = cdiv(s.m, TN_T);
    P.tiles_n = cdiv(s.n, TN_T);
    // equal chunks of at most TN_KC rows, multiples of 4 (the MFMA k-step)
    int split = cdiv(s.k, KC);
    const int kps = round_up(cdiv(s.k, split), 4);
    split = cdiv(s.k, kps);
    P.k_per_split = kps;
    P.block_begin = total;
    g.begins[i] = total;
    total += P.tiles_m * P.tiles_n * s.batch * split;
  }
  for (int i = count; i < MFM_TN_MAXP; ++i) g.begins[i] = 0x7fffffff;
  const size_t lds = (size_t)2 * KC * TN_T * sizeof(float);
#define MFM_TN_GO(KC_)                                                                                             \
  do {                                                                                                             \
    auto* fn = gemm_tn_kernel<KC_>;                                                                                \
    if (lds > 64 * 1024) MFM_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    MFM_LAUNCH_TIMED(fn, dim3(total), dim3(256), lds, stream, g);                                                \
  } while (0)
  if (KC == 80) MFM_TN_GO(80);
  else if (KC == 320) MFM_TN_GO(320);
  else MFM_TN_GO(160);
#undef MFM_TN_GO
  MFM_LAUNCH_CHECK("gemm_tn_kernel");
  return MFM_OK;
}

}  // namespace mfm
