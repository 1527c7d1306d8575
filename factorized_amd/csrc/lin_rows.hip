// Row-block linear layers for the Memory Fusion Network's attention block at small T*B (round 2):
//   forward   C[M, n] = epi(A[M, k] W[n, k]^T + bias)          M = T*B rows, k <= 512, fp32
//   backward  C[M, n] (+)= epi(A[M, k] Wt[k, n])               (trans: Wt is the same row-major weight, read along its rows)
// The grouped GEMM (gemm.hip) walks K in 32-deep steps through a register ring; with one or three such products per
// launch and M = 640 rows there are only 80-260 tiles, and each tile's K loop is a chain of 13 dependent load round
// trips (16-17 us per launch, four of them in the forward).  Here a workgroup owns 16 rows x 32 output columns and
// requests EVERYTHING it will read -- its 16 x k slice of A and its 32 x k slice of W, up to 24 16-byte loads per thread
// -- before it touches any of it: one memory round trip, then the four waves split the k range, run their share of the
// v_mfma_f32_16x16x4_f32 steps out of LDS and add their partial tiles through LDS.  Epilogues as in the grouped GEMM
// (internal.h::GemmEpi kinds 0-2; the dropout stream is the same function of (seed, op_id, row, col)).
#include <stdlib.h>

#include <algorithm>

#include "internal.h"

namespace mfm {

namespace {

constexpr int LR_ROWS = 16, LR_COLS = 32, LR_THREADS = 256;

struct LinRowsDevItem {
  const float* a; const float* w; const float* bias; float* c; float* aux;
  int lda, ldw, ldc, n, k, kind, block_begin, col_groups, trans, accumulate;
  float p; unsigned op_id;
};
struct LinRowsDev {
  LinRowsDevItem it[MFM_LINROWS_MAX];
  int count, M, train;
  unsigned long long seed;
  const unsigned long long* tick;
};

// PA / PWT: 16-byte loads per thread for the A slice (16 x k) and the W slice (32 x k)
template <int PA, int PWT>
__global__ __launch_bounds__(LR_THREADS) void lin_rows_kernel(const LinRowsDev L) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int bi = lane & 15, q = lane >> 4;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < MFM_LINROWS_MAX; ++i) pi += (i < L.count && (int)blockIdx.x >= L.it[i].block_begin) ? 1 : 0;
  const LinRowsDevItem& I = L.it[pi];
  const int local = (int)blockIdx.x - I.block_begin;
  const int cg = local % I.col_groups, rt = local / I.col_groups;
  const int row0 = rt * LR_ROWS, c0 = cg * LR_COLS;
  const int K = I.k, K4 = K >> 2, LD = K + 4;
  float* As = lds;                       // [16][K + 4]
  float* Ws = lds + LR_ROWS * LD;        // [32][K + 4]

  // ---- every operand byte of this workgroup is requested here, before anything waits
  const unsigned inv = ((1u << 20) + K4 - 1) / K4;          // idx / K4 == (idx * inv) >> 20 for idx < 4096, K4 <= 128
  const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc((void*)I.a, 0, (int)(((int64_t)(L.M - 1) * I.lda + K) * 4), 0x00020000);
  const int w_elems = I.trans ? (K - 1) * I.ldw + I.n : (I.n - 1) * I.ldw + K;
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)I.w, 0, w_elems * 4, 0x00020000);
  f32x4 ra[PA], rw[PWT];
#pragma unroll
  for (int j = 0; j < PA; ++j) {
    const int idx = tid + j * LR_THREADS;
    const int r = (int)(((unsigned)idx * inv) >> 20), k4 = idx - r * K4;
    const bool ok = (idx < LR_ROWS * K4) & (row0 + r < L.M);
    const int off = ok ? ((row0 + r) * I.lda + 4 * k4) * 4 : -16;       // out of range: the load returns zeros
    ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ares, off, 0, 0));
  }
  // W slice.  Forward: rows c0 .. c0+31 of W[n, k], k-contiguous, image [32][K + 4].  trans: columns c0 .. c0+31 of
  // Wt[k, n], eight 16-byte groups per k row, image [K][32] with the columns of odd k rows rotated by 16 so that the two
  // 32-lane halves of a B-fragment read (k rows 4 ks + {0, 1} / {2, 3}, 16 columns each) touch 32 distinct banks.
  // Groups that run past column n read the next row (in range of the buffer): those outputs are never stored.
  const bool trans = I.trans != 0;
#pragma unroll
  for (int j = 0; j < PWT; ++j) {
    const int idx = tid + j * LR_THREADS;
    int off;
    if (trans) {
      const int r = idx >> 3, c4 = idx & 7;
      const bool ok = (r < K) & (c0 + 4 * c4 < I.n);
      off = ok ? (r * I.ldw + c0 + 4 * c4) * 4 : -16;
    } else {
      const int r = (int)(((unsigned)idx * inv) >> 20), k4 = idx - r * K4;
      const bool ok = (idx < LR_COLS * K4) & (c0 + r < I.n);
      off = ok ? ((c0 + r) * I.ldw + 4 * k4) * 4 : -16;
    }
    rw[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, off, 0, 0));
  }
#pragma unroll
  for (int j = 0; j < PA; ++j) {
    const int idx = tid + j * LR_THREADS;
    const int r = (int)(((unsigned)idx * inv) >> 20), k4 = idx - r * K4;
    if (idx < LR_ROWS * K4) *reinterpret_cast<f32x4*>(As + r * LD + 4 * k4) = ra[j];
  }
#pragma unroll
  for (int j = 0; j < PWT; ++j) {
    const int idx = tid + j * LR_THREADS;
    if (trans) {
      const int r = idx >> 3, c4 = idx & 7;
      if (r < K) *reinterpret_cast<f32x4*>(Ws + r * LR_COLS + ((4 * c4 + 16 * (r & 1)) & 31)) = rw[j];
    } else {
      const int r = (int)(((unsigned)idx * inv) >> 20), k4 = idx - r * K4;
      if (idx < LR_COLS * K4) *reinterpret_cast<f32x4*>(Ws + r * LD + 4 * k4) = rw[j];
    }
  }
  __syncthreads();

  // ---- wave w takes the k-steps w, w+4, ...: two 16x16 output fragments, partial over its share of k
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  {
    // element (k = 4 ks + q, column f 16 + bi) of the W image: forward [column][k], trans [k][column rotated by 16 (k & 1)]
    const float* ap = As + bi * LD + q;
    const int ws = trans ? 4 * LR_COLS : 4;                            // stride of one k-step
    const float* w0 = trans ? Ws + q * LR_COLS + ((bi + 16 * (q & 1)) & 31) : Ws + bi * LD + q;
    const float* w1 = trans ? Ws + q * LR_COLS + ((bi + 16 + 16 * (q & 1)) & 31) : Ws + (16 + bi) * LD + q;
    const int nks = (K4 - wave + 3) >> 2;
    int i = 0;
    for (; i + 4 <= nks; i += 4) {         // four k-steps' operands are read before their MFMAs
      float a[4], b0[4], b1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ks = 4 * (i + u) + wave;
        a[u] = ap[4 * ks]; b0[u] = w0[ws * ks]; b1[u] = w1[ws * ks];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { acc0 = mma16x16x4(a[u], b0[u], acc0); acc1 = mma16x16x4(a[u], b1[u], acc1); }
    }
    for (; i < nks; ++i) {
      const int ks = 4 * i + wave;
      acc0 = mma16x16x4(ap[4 * ks], w0[ws * ks], acc0);
      acc1 = mma16x16x4(ap[4 * ks], w1[ws * ks], acc1);
    }
  }
  __syncthreads();                       // the operand images are dead: their LDS holds the partial tiles now
  f32x4* red = reinterpret_cast<f32x4*>(lds);        // [wave][fragment][lane]
  red[(wave * 2 + 0) * 64 + lane] = acc0;
  red[(wave * 2 + 1) * 64 + lane] = acc1;
  __syncthreads();

  // ---- epilogue: thread t finishes row t >> 4, columns (t & 15) and 16 + (t & 15): 64-byte runs per 16 threads
  const int ri = tid >> 4, cj = tid & 15;
  const int rl = (ri >> 2) * 16 + cj, rr = ri & 3;          // accumulator lane / element of (row ri, column cj)
  const int row = row0 + ri;
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int col = c0 + 16 * f + cj;
    float v = 0.0f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += reinterpret_cast<const float*>(red + (w * 2 + f) * 64 + rl)[rr];
    if (row < L.M && col < I.n) {
      if (I.bias) v += I.bias[col];
      const int64_t off = (int64_t)row * I.ldc + col;
      if (I.kind == 1) {                 // relu + dropout, mask kept for the backward (GemmEpi kind 1)
        float mk = 1.0f;
        if (L.train && I.p > 0.0f) {
          const uint64_t idx = ((uint64_t)I.op_id << 40) + (uint64_t)row * (uint64_t)I.n + (uint64_t)col;
          mk = (rng_uniform(L.seed + (L.tick ? *L.tick : 0ull), idx) < I.p) ? 0.0f : 1.0f / (1.0f - I.p);
        }
        I.aux[off] = (v > 0.0f) ? mk : 0.0f;
        v = fmaxf(v, 0.0f) * mk;
      } else if (I.kind == 2) {
        v = act_tanh(v);
      } else if (I.kind == 3) {          // backward of kind 1: times the saved relu / dropout mask
        v *= I.aux[off];
      }
      if (I.accumulate) atomicAdd(I.c + off, v);      // several products into one zero-filled / pre-filled output
      else I.c[off] = v;
    }
  }
}

}  // namespace

bool lin_rows_supported(const LinRowsItem* items, int count, int M) {
  if (count < 1 || count > MFM_LINROWS_MAX || M < 1) return false;
  for (int i = 0; i < count; ++i) {
    const LinRowsItem& it = items[i];
    if (!it.a || !it.w || !it.c || it.n < 1 || it.k < 4 || it.k > 512 || (it.k & 3)) return false;
    if (it.kind < 0 || it.kind > 3 || ((it.kind == 1 || it.kind == 3) && !it.aux)) return false;
    if (it.accumulate && it.kind != 0) return false;
    const int64_t we = it.trans ? (int64_t)(it.k - 1) * it.ldw + it.n : (int64_t)(it.n - 1) * it.ldw + it.k;
    if ((int64_t)(M - 1) * it.lda + it.k >= ((int64_t)1 << 29) || we >= ((int64_t)1 << 29)) return false;
  }
  return true;
}

int lin_rows_launch(const LinRowsItem* items, int count, int M, int train, unsigned long long seed, hipStream_t stream,
                    const unsigned long long* tick) {
  MFM_REQUIRE(lin_rows_supported(items, count, M), "lin_rows: unsupported shapes (count %d, M %d)", count, M);
  LinRowsDev L;
  memset(&L, 0, sizeof(L));
  L.count = count; L.M = M; L.train = train; L.seed = seed; L.tick = tick;
  const int row_tiles = cdiv(M, LR_ROWS);
  int total = 0, kmax = 0;
  size_t img = 0;
  for (int i = 0; i < count; ++i) {
    const LinRowsItem& s = items[i];
    LinRowsDevItem& d = L.it[i];
    d.a = s.a; d.w = s.w; d.bias = s.bias; d.c = s.c; d.aux = s.aux;
    d.lda = s.lda; d.ldw = s.ldw; d.ldc = s.ldc; d.n = s.n; d.k = s.k; d.kind = s.kind; d.p = s.p; d.op_id = s.op_id; d.trans = s.trans; d.accumulate = s.accumulate;
    d.col_groups = cdiv(s.n, LR_COLS);
    d.block_begin = total;
    total += row_tiles * d.col_groups;
    kmax = std::max(kmax, s.k);
    img = std::max(img, (size_t)LR_ROWS * (s.k + 4) + (s.trans ? (size_t)s.k * LR_COLS : (size_t)LR_COLS * (s.k + 4)));
  }
  // one instantiation per launch: the per-thread load counts follow the longest k of the group
  // operand images, reused for the four waves' partial tiles (4 x 2 x 64 lanes x 16 bytes)
  const size_t lds = std::max(img * sizeof(float), (size_t)4 * 2 * 64 * 16);
  const int k4 = kmax >> 2;
#define MFM_LR_GO(PA_, PW_)                                                                                       \
  do {                                                                                                            \
    auto* fn = lin_rows_kernel<PA_, PW_>;                                                                         \
    if (lds > 64 * 1024) MFM_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    MFM_LAUNCH_TIMED(fn, dim3(total), dim3(LR_THREADS), lds, stream, L);                                        \
  } while (0)
  if (k4 <= 32) MFM_LR_GO(2, 4);
  else if (k4 <= 64) MFM_LR_GO(4, 8);
  else MFM_LR_GO(8, 16);
#undef MFM_LR_GO
  MFM_LAUNCH_CHECK("lin_rows_kernel");
  return MFM_OK;
}

}  // namespace mfm
