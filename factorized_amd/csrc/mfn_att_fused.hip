// Memory Fusion Network attention block, one launch per direction (fp32 plans at small T*B).
//
// Reference (mfm_model.py:171-176, evaluated for all T steps at once -- see mfn_att.hip):
//   cStar      = [c_{t-1} (l,a,v) , c_t (l,a,v)]
//   attention  = softmax(att1_fc2(drop(relu(att1_fc1(cStar)))))
//   attended   = attention * cStar
//   cHat       = tanh(att2_fc2(drop(relu(att2_fc1(attended)))))
//   a_n        = gamma_n_fc1[:, :A2] attended + b                 (the memory columns are applied in mfn_mem.hip)
//
// As grouped GEMMs + row kernels this chain is 6 launches forward and 6 backward; at the reference's B=32 (T*B = 640
// rows) each costs ~8-10 us of which almost all is launch boundary and cold first fetches.  Here a workgroup owns 16
// (t, b) rows and walks the whole chain with every intermediate in LDS; the saved tensors the weight-gradient GEMMs and
// the memory recurrence need are written out as they are produced.  Per tile the chain is 4.1 k fp32 MFMAs (~14 us of
// matrix time on one CU, 40 tiles at T*B = 640).
//
// STATUS: parity-green on every MFN test (tests/test_gpu_mfn_plan.py runs both forms), but SLOWER than the launches it
// replaces: 61 us forward against ~46 us (profiles/r02_mfn_att_fused.txt has the per-phase clock readings, build with
// MFM_EXTRA_FLAGS=-DMFM_ATT_TIMING).  Each tile streams the block's 1 MB of weights through one CU while 5 tiles per
// XCD miss on the same cold lines (the optimizer rewrote them a step ago): ~2 us per dependent load, and the row-wise
// passes between the products pay that per loop iteration.  It is therefore opt-in (MFM_MFN_FUSED=1); the plan default
// is the GEMM form.  What it would take: the elementwise passes' operands requested in one batch, weight panels staged
// through LDS by all 512 threads two panels ahead, and the 16-row tiles split over more CUs.
//
// 512 threads = 8 waves, MFMA 16x16x4 fp32.  Output fragments (16 columns) go round-robin over the waves.
//   y = x W^T (LinF):  lane (bi, q) takes units 16 j + 4 q + {0..3} of row bi of x (LDS, one 16-byte read) and of row
//                      n = bi of W (one 16-byte buffer load) per 16-wide block j -> four MFMAs.
//   dx = dy W (LinB):  lane (bi, q) takes W[4 ks + q][16 f + bi] (dword buffer load) per reduction step ks.
// Weight requests run AT_D / AT_DB units ahead in a register ring; the first ring of a product is requested before the
// barrier / elementwise pass that precedes it.
#include <hip/hip_runtime.h>
#include "internal.h"

namespace mfm {

namespace {

constexpr int AT_THREADS = 512;
constexpr int AT_WAVES = 8;
constexpr int AT_ROWS = 16;
constexpr int AT_D = 16;                     // LinF ring: 16 x 16 bytes per lane, 4 MFMAs per slot
constexpr int AT_DB = 48;                    // LinB ring: 48 dwords per lane, 1 MFMA per slot
constexpr int AT_OOB = 0x7FFFFFF0;

__host__ __device__ inline int at_ld(int n) { return ((n + 63) & ~63) + 4; }      // (ld / 4) odd, ld == 4 (mod 64)
__device__ __forceinline__ int pad16(int n) { return (n + 15) & ~15; }

struct Ctx { int tid, lane, wave, bi, q; };

#ifdef MFM_ATT_TIMING
#define ATT_MARK(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) ts_[i] = wall_clock64(); } while (0)
#define ATT_MARK_DECL long long ts_[16] = {0}
#define ATT_MARK_PRINT(n, tag) do { if (threadIdx.x == 0 && blockIdx.x == 0) { printf("%s", tag); for (int i_ = 1; i_ < n; ++i_) printf(" %lld", ts_[i_] - ts_[i_ - 1]); printf("  (10 ns ticks)\n"); } } while (0)
#else
#define ATT_MARK(i)
#define ATT_MARK_DECL
#define ATT_MARK_PRINT(n, tag)
#endif

// ---- y[16, N] = x[16, K] W^T,  W [N, K] with row stride ldw (elements past K in a row are finite: x is zero there)
struct LinF {
  __amdgpu_buffer_rsrc_t res;
  int ldw, N, J, U;
  int rfi, rj, ru;
  f32x4 ring[AT_D];
  __device__ __forceinline__ f32x4 request(const Ctx& c) {
    const int n = (c.wave + AT_WAVES * rfi) * 16 + c.bi;
    const bool ok = (int)(ru < U) & (int)(n < N);
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res, ok ? (n * ldw + 16 * rj + 4 * c.q) * 4 : AT_OOB, 0, 0));
    ++ru;
    if (++rj == J) { rj = 0; ++rfi; }
    return v;
  }
  __device__ __forceinline__ void prime(const Ctx& c, const float* w, int ldw_, int N_, int K) {
    res = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, N_ * ldw_ * 4, 0x00020000);
    ldw = ldw_; N = N_; J = pad16(K) >> 4;
    const int NF = (N_ + 15) >> 4;
    const int nfr = NF > c.wave ? (NF - c.wave + AT_WAVES - 1) / AT_WAVES : 0;
    U = nfr * J;
    rfi = rj = ru = 0;
#pragma unroll
    for (int i = 0; i < AT_D; ++i) ring[i] = request(c);
  }
  // epi(frag, acc) once per finished output fragment
  template <typename Epi>
  __device__ __forceinline__ void run(const Ctx& c, const float* X, int ldx, Epi epi) {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    int fi = 0, j = 0;
    for (int u0 = 0; u0 < U; u0 += AT_D) {
#pragma unroll
      for (int i = 0; i < AT_D; ++i) {
        if (u0 + i < U) {
          const f32x4 hv = *reinterpret_cast<const f32x4*>(X + c.bi * ldx + 16 * j + 4 * c.q);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = mma16x16x4(hv[e], ring[i][e], acc);
          if (++j == J) {
            epi(c.wave + AT_WAVES * fi, acc);
            acc = f32x4{0.f, 0.f, 0.f, 0.f};
            j = 0; ++fi;
          }
        }
        ring[i] = request(c);
      }
    }
  }
};

// ---- dx[16, N] = dy[16, K] W,  W [K, N'] with row stride ldw (only columns < N are used)
struct LinB {
  __amdgpu_buffer_rsrc_t res;
  int ldw, N, K, KS, U;
  int rfi, rks, ru;
  float ring[AT_DB];
  __device__ __forceinline__ float request(const Ctx& c) {
    const int col = (c.wave + AT_WAVES * rfi) * 16 + c.bi, k = 4 * rks + c.q;
    const bool ok = (int)(ru < U) & (int)(col < N) & (int)(k < K);
    const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(res, ok ? (k * ldw + col) * 4 : AT_OOB, 0, 0));
    ++ru;
    if (++rks == KS) { rks = 0; ++rfi; }
    return v;
  }
  __device__ __forceinline__ void prime(const Ctx& c, const float* w, int ldw_, int N_, int K_) {
    res = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, K_ * ldw_ * 4, 0x00020000);
    ldw = ldw_; N = N_; K = K_; KS = (K_ + 3) >> 2;
    const int NF = (N_ + 15) >> 4;
    const int nfr = NF > c.wave ? (NF - c.wave + AT_WAVES - 1) / AT_WAVES : 0;
    U = nfr * KS;
    rfi = rks = ru = 0;
#pragma unroll
    for (int i = 0; i < AT_DB; ++i) ring[i] = request(c);
  }
  template <typename Epi>
  __device__ __forceinline__ void run(const Ctx& c, const float* X, int ldx, Epi epi) {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    int fi = 0, ks = 0;
    for (int u0 = 0; u0 < U; u0 += AT_DB) {
#pragma unroll
      for (int i = 0; i < AT_DB; ++i) {
        if (u0 + i < U) {
          const float a = X[c.bi * ldx + 4 * ks + c.q];          // columns >= K of the tile are zero
          acc = mma16x16x4(a, ring[i], acc);
          if (++ks == KS) {
            epi(c.wave + AT_WAVES * fi, acc);
            acc = f32x4{0.f, 0.f, 0.f, 0.f};
            ks = 0; ++fi;
          }
        }
        ring[i] = request(c);
      }
    }
  }
};

// raw accumulator tile -> LDS tile (row = 4 q + r, col = 16 frag + bi), plain or accumulating
struct ToLds {
  float* Y; int ld; const Ctx* c; bool add;
  __device__ __forceinline__ void operator()(int frag, const f32x4& acc) const {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* p = Y + (4 * c->q + r) * ld + frag * 16 + c->bi;
      *p = add ? *p + acc[r] : acc[r];
    }
  }
};
// raw accumulator tile + bias (optionally tanh) -> global [rows, N]
struct ToGlobal {
  float* out; const float* bias; int N, row0, rows; const Ctx* c; bool tanh_;
  __device__ __forceinline__ void operator()(int frag, const f32x4& acc) const {
    const int col = frag * 16 + c->bi;
    if (col >= N) return;
    const float bv = bias[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + 4 * c->q + r;
      if (row < rows) {
        const float v = acc[r] + bv;
        out[(int64_t)row * N + col] = tanh_ ? act_tanh(v) : v;
      }
    }
  }
};

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_add(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// tile[r][c] for c in [n, ld): zeros (reduction padding of the products that read the tile)
__device__ __forceinline__ void zero_tail(float* tile, int ld, int n, const Ctx& c) {
  const int w = ld - n;
  for (int idx = c.tid; idx < AT_ROWS * w; idx += AT_THREADS) {
    const int r = idx / w, k = idx - r * w;
    tile[r * ld + n + k] = 0.0f;
  }
}

// relu + dropout of a raw tile, in place; the kept-scale mask and the activation go to global (same stream as the GEMM
// epilogue kind 1: gemm_common.h)
__device__ __forceinline__ void relu_drop(float* tile, int ld, int n, const float* bias, float* act, float* mask, float p,
                                          int train, unsigned long long seed, unsigned op_id, int row0, int rows, const Ctx& c) {
  for (int idx = c.tid; idx < AT_ROWS * n; idx += AT_THREADS) {
    const int r = idx / n, col = idx - r * n;
    const int row = row0 + r;
    float v = tile[r * ld + col] + bias[col];
    float mk = 1.0f;
    if (train && p > 0.0f) {
      const uint64_t id = ((uint64_t)op_id << 40) + (uint64_t)row * (uint64_t)n + (uint64_t)col;
      mk = (rng_uniform(seed, id) < p) ? 0.0f : 1.0f / (1.0f - p);
    }
    const float keep = (v > 0.0f) ? mk : 0.0f;
    v = fmaxf(v, 0.0f) * mk;
    if (row >= rows) v = 0.0f;
    tile[r * ld + col] = v;
    if (row < rows) {
      act[(int64_t)row * n + col] = v;
      mask[(int64_t)row * n + col] = keep;
    }
  }
}

__global__ __launch_bounds__(AT_THREADS) void mfn_att_fwd_kernel(const MfnAttFused L) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  Ctx c;
  c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = c.tid >> 6; c.bi = c.lane & 15; c.q = c.lane >> 4;
  const int A2 = L.A2, ldA = at_ld(A2), ldH = at_ld(max(max(L.nn1, L.nn2), 1));
  float* Xc = sm;                            // cStar          [16][ldA]
  float* Lg = Xc + AT_ROWS * ldA;            // logits -> attention
  float* At = Lg + AT_ROWS * ldA;            // attended
  float* H1 = At + AT_ROWS * ldA;            // [16][ldH]
  float* H2 = H1 + AT_ROWS * ldH;
  const int row0 = blockIdx.x * AT_ROWS, rows = L.T * L.B;

  ATT_MARK_DECL;
  ATT_MARK(0);
  LinF g1;
  g1.prime(c, L.w_att1_1, A2, L.nn1, A2);
  // ---- cStar rows: gathered from the three LSTMs' cell states (c_{-1} = 0), kept in LDS and saved
  for (int idx = c.tid; idx < AT_ROWS * A2; idx += AT_THREADS) {
    const int r = idx / A2, col = idx - r * A2;
    const int row = row0 + r;
    float v = 0.0f;
    if (row < rows) {
      const int half = col >= L.tot;
      const int cc = half ? col - L.tot : col;
      const int m = (cc >= L.off[1]) + (cc >= L.off[2]);
      const int t = row / L.B, b = row - t * L.B;
      const int ts = half ? t : t - 1;
      if (ts >= 0) v = L.cs[m][((int64_t)ts * L.B + b) * L.Hp[m] + cc - L.off[m]];
      L.cstar[(int64_t)row * A2 + col] = v;
    }
    Xc[r * ldA + col] = v;
  }
  zero_tail(Xc, ldA, A2, c);
  lds_barrier();
  ATT_MARK(1);

  // ---- h1 = drop(relu(att1_fc1(cStar)))
  g1.run(c, Xc, ldA, ToLds{H1, ldH, &c, false});
  ATT_MARK(2);
  LinF g2;
  g2.prime(c, L.w_att1_2, L.nn1, A2, L.nn1);
  lds_barrier();
  relu_drop(H1, ldH, L.nn1, L.b_att1_1, L.h1, L.m1, L.p1, L.train, L.seed, 101u, row0, rows, c);
  zero_tail(H1, ldH, L.nn1, c);
  lds_barrier();
  ATT_MARK(3);

  // ---- attention = softmax(att1_fc2(h1)), attended = attention * cStar
  g2.run(c, H1, ldH, ToLds{Lg, ldA, &c, false});
  ATT_MARK(4);
  LinF g3;
  g3.prime(c, L.w_att2_1, A2, L.nn2, A2);
  lds_barrier();
  for (int rr = 0; rr < AT_ROWS / AT_WAVES; ++rr) {
    const int r = c.wave * (AT_ROWS / AT_WAVES) + rr;
    const int row = row0 + r;
    float mx = -3.0e38f;
    for (int col = c.lane; col < A2; col += 64) {
      const float v = Lg[r * ldA + col] + L.b_att1_2[col];
      Lg[r * ldA + col] = v;
      mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    float s = 0.0f;
    for (int col = c.lane; col < A2; col += 64) {
      const float e = expf(Lg[r * ldA + col] - mx);
      Lg[r * ldA + col] = e;
      s += e;
    }
    const float inv = 1.0f / wave_add(s);
    for (int col = c.lane; col < A2; col += 64) {
      const float p = Lg[r * ldA + col] * inv;
      const float a = (row < rows) ? p * Xc[r * ldA + col] : 0.0f;
      At[r * ldA + col] = a;
      if (row < rows) {
        L.att[(int64_t)row * A2 + col] = p;
        L.attended[(int64_t)row * A2 + col] = a;
      }
    }
  }
  zero_tail(At, ldA, A2, c);
  lds_barrier();
  ATT_MARK(5);

  // ---- h2 = drop(relu(att2_fc1(attended))) ; a_n = gamma_n_fc1[:, :A2] attended + b
  g3.run(c, At, ldA, ToLds{H2, ldH, &c, false});
  ATT_MARK(6);
  {
    LinF ga;
    ga.prime(c, L.w_gam1, A2 + L.M, L.g1, A2);
    ga.run(c, At, ldA, ToGlobal{L.a1, L.b_gam1, L.g1, row0, rows, &c, false});
    ga.prime(c, L.w_gam2, A2 + L.M, L.g2, A2);
    ga.run(c, At, ldA, ToGlobal{L.a2, L.b_gam2, L.g2, row0, rows, &c, false});
  }
  ATT_MARK(7);
  LinF g4;
  g4.prime(c, L.w_att2_2, L.nn2, L.M, L.nn2);
  lds_barrier();
  relu_drop(H2, ldH, L.nn2, L.b_att2_1, L.h2, L.m2, L.p2, L.train, L.seed, 102u, row0, rows, c);
  zero_tail(H2, ldH, L.nn2, c);
  lds_barrier();
  ATT_MARK(8);
  // ---- cHat = tanh(att2_fc2(h2))
  g4.run(c, H2, ldH, ToGlobal{L.chat, L.b_att2_2, L.M, row0, rows, &c, true});
  ATT_MARK(9);
  ATT_MARK_PRINT(10, "att fwd: gather g1 relu1 g2 softmax g3 ga12 relu2 g4:");
}

// Backward of the block: from d(pre-tanh cHat), du_1, du_2 (memory recurrence BPTT) to d cStar scattered onto the three
// LSTMs' cell-state gradients (atomic adds of at most two terms per element into a buffer the step's first launch
// cleared: commutative, so deterministic), leaving dh2, d logits, dh1 for the weight-gradient GEMMs.
__global__ __launch_bounds__(AT_THREADS) void mfn_att_bwd_kernel(const MfnAttFused L) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  Ctx c;
  c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = c.tid >> 6; c.bi = c.lane & 15; c.q = c.lane >> 4;
  const int A2 = L.A2, ldA = at_ld(A2), ldM = at_ld(L.M);
  const int ld1 = at_ld(L.nn1), ld2 = at_ld(L.nn2), ldg1 = at_ld(L.g1), ldg2 = at_ld(L.g2);
  float* DA = sm;                            // d attended      [16][ldA]
  float* DL = DA + AT_ROWS * ldA;            // d logits
  float* DC = DL + AT_ROWS * ldA;            // d cStar
  float* D4 = DC + AT_ROWS * ldA;            // d pre-tanh cHat [16][ldM]
  float* U1 = D4 + AT_ROWS * ldM;            // du_1            [16][ldg1]
  float* U2 = U1 + AT_ROWS * ldg1;
  float* DH2 = U2 + AT_ROWS * ldg2;          // [16][ld2]
  float* DH1 = DH2 + AT_ROWS * ld2;          // [16][ld1]
  const int row0 = blockIdx.x * AT_ROWS, rows = L.T * L.B;

  LinB b1;
  b1.prime(c, L.w_att2_2, L.nn2, L.nn2, L.M);
  auto load_tile = [&](float* tile, int ld, const float* src, int n) {
    for (int idx = c.tid; idx < AT_ROWS * ld; idx += AT_THREADS) {
      const int r = idx / ld, col = idx - r * ld;
      const int row = row0 + r;
      tile[idx] = (col < n && row < rows) ? src[(int64_t)row * n + col] : 0.0f;
    }
  };
  load_tile(D4, ldM, L.dchat, L.M);
  load_tile(U1, ldg1, L.du1, L.g1);
  load_tile(U2, ldg2, L.du2, L.g2);
  lds_barrier();

  // ---- dh2 = (d pre-cHat W_att2_fc2) * mask2
  b1.run(c, D4, ldM, ToLds{DH2, ld2, &c, false});
  LinB b2;
  b2.prime(c, L.w_att2_1, A2, A2, L.nn2);
  lds_barrier();
  for (int idx = c.tid; idx < AT_ROWS * ld2; idx += AT_THREADS) {
    const int r = idx / ld2, col = idx - r * ld2;
    const int row = row0 + r;
    float v = 0.0f;
    if (col < L.nn2 && row < rows) {
      v = DH2[idx] * L.m2[(int64_t)row * L.nn2 + col];
      L.dh2[(int64_t)row * L.nn2 + col] = v;
    }
    DH2[idx] = v;
  }
  lds_barrier();

  // ---- d attended = dh2 W_att2_fc1 + du1 W_gamma1_fc1[:, :A2] + du2 W_gamma2_fc1[:, :A2]
  b2.run(c, DH2, ld2, ToLds{DA, ldA, &c, false});
  b2.prime(c, L.w_gam1, A2 + L.M, A2, L.g1);
  b2.run(c, U1, ldg1, ToLds{DA, ldA, &c, true});
  b2.prime(c, L.w_gam2, A2 + L.M, A2, L.g2);
  b2.run(c, U2, ldg2, ToLds{DA, ldA, &c, true});
  LinB b3;
  b3.prime(c, L.w_att1_2, L.nn1, L.nn1, A2);
  lds_barrier();

  // ---- through attended = attention * cStar and the softmax
  for (int rr = 0; rr < AT_ROWS / AT_WAVES; ++rr) {
    const int r = c.wave * (AT_ROWS / AT_WAVES) + rr;
    const int row = row0 + r;
    float s = 0.0f;
    for (int col = c.lane; col < A2; col += 64) {
      float p = 0.0f, cs = 0.0f;
      if (row < rows) { p = L.att[(int64_t)row * A2 + col]; cs = L.cstar[(int64_t)row * A2 + col]; }
      const float da = DA[r * ldA + col];
      const float g = da * cs;
      s += g * p;
      DC[r * ldA + col] = da * p;
      DL[r * ldA + col] = g;
      DA[r * ldA + col] = p;
    }
    s = wave_add(s);
    for (int col = c.lane; col < A2; col += 64) {
      const float v = DA[r * ldA + col] * (DL[r * ldA + col] - s);
      DL[r * ldA + col] = v;
      if (row < rows) L.dlog[(int64_t)row * A2 + col] = v;
    }
  }
  zero_tail(DL, ldA, A2, c);
  lds_barrier();

  // ---- dh1 = (d logits W_att1_fc2) * mask1
  b3.run(c, DL, ldA, ToLds{DH1, ld1, &c, false});
  LinB b4;
  b4.prime(c, L.w_att1_1, A2, A2, L.nn1);
  lds_barrier();
  for (int idx = c.tid; idx < AT_ROWS * ld1; idx += AT_THREADS) {
    const int r = idx / ld1, col = idx - r * ld1;
    const int row = row0 + r;
    float v = 0.0f;
    if (col < L.nn1 && row < rows) {
      v = DH1[idx] * L.m1[(int64_t)row * L.nn1 + col];
      L.dh1[(int64_t)row * L.nn1 + col] = v;
    }
    DH1[idx] = v;
  }
  lds_barrier();

  // ---- d cStar = d attended * attention + dh1 W_att1_fc1, scattered: dc_t += d cStar_t[second half] + d cStar_{t+1}[first half]
  b4.run(c, DH1, ld1, ToLds{DC, ldA, &c, true});
  lds_barrier();
  for (int idx = c.tid; idx < AT_ROWS * A2; idx += AT_THREADS) {
    const int r = idx / A2, col = idx - r * A2;
    const int row = row0 + r;
    if (row >= rows) continue;
    const int half = col >= L.tot;
    const int cc = half ? col - L.tot : col;
    const int m = (cc >= L.off[1]) + (cc >= L.off[2]);
    const int t = row / L.B, b = row - t * L.B;
    const int ts = half ? t : t - 1;
    if (ts >= 0) atomicAdd(L.dcx[m] + ((int64_t)ts * L.B + b) * L.Hp[m] + cc - L.off[m], DC[r * ldA + col]);
  }
}

size_t fwd_lds(const MfnAttFused& L) {
  return ((size_t)3 * AT_ROWS * at_ld(L.A2) + (size_t)2 * AT_ROWS * at_ld(std::max(L.nn1, L.nn2))) * sizeof(float);
}
size_t bwd_lds(const MfnAttFused& L) {
  return ((size_t)3 * AT_ROWS * at_ld(L.A2) +
          (size_t)AT_ROWS * (at_ld(L.M) + at_ld(L.g1) + at_ld(L.g2) + at_ld(L.nn2) + at_ld(L.nn1))) * sizeof(float);
}
constexpr size_t AT_LDS_MAX = 150 * 1024;

}  // namespace

bool mfn_att_fused_supported(const MfnAttFused& L) {
  const int64_t big = (int64_t)1 << 28;
  return fwd_lds(L) <= AT_LDS_MAX && bwd_lds(L) <= AT_LDS_MAX && (int64_t)L.A2 * (L.A2 + L.M) < big &&
         (int64_t)std::max(L.g1, L.g2) * (L.A2 + L.M) < big && (int64_t)L.T * L.B * L.A2 < ((int64_t)1 << 31);
}

static int set_lds_attrs() {
  static bool attr = false;
  if (!attr) {
    MFM_HIP_CHECK(hipFuncSetAttribute((const void*)mfn_att_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AT_LDS_MAX));
    MFM_HIP_CHECK(hipFuncSetAttribute((const void*)mfn_att_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AT_LDS_MAX));
    attr = true;
  }
  return MFM_OK;
}

int mfn_att_fused_fwd_launch(const MfnAttFused& L, hipStream_t stream) {
  MFM_REQUIRE(mfn_att_fused_supported(L), "mfn fused attention: unsupported sizes");
  if (int rc = set_lds_attrs()) return rc;
  const int tiles = (L.T * L.B + AT_ROWS - 1) / AT_ROWS;
  hipLaunchKernelGGL(mfn_att_fwd_kernel, dim3(tiles), dim3(AT_THREADS), fwd_lds(L), stream, L);
  MFM_LAUNCH_CHECK("mfn_att_fwd_kernel");
  return MFM_OK;
}

int mfn_att_fused_bwd_launch(const MfnAttFused& L, hipStream_t stream) {
  MFM_REQUIRE(mfn_att_fused_supported(L), "mfn fused attention: unsupported sizes");
  if (int rc = set_lds_attrs()) return rc;
  const int tiles = (L.T * L.B + AT_ROWS - 1) / AT_ROWS;
  hipLaunchKernelGGL(mfn_att_bwd_kernel, dim3(tiles), dim3(AT_THREADS), bwd_lds(L), stream, L);
  MFM_LAUNCH_CHECK("mfn_att_bwd_kernel");
  return MFM_OK;
}

}  // namespace mfm
