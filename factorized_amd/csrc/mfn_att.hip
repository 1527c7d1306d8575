// Memory Fusion Network, the part of MFN.forward that does NOT depend on the memory (reference mfm_model.py:171-176):
//   cStar_t    = [c_{t-1} (l,a,v) , c_t (l,a,v)]                           (:171-173)
//   attention  = softmax(att1_fc2(drop(relu(att1_fc1(cStar_t)))))           (:174)
//   attended_t = attention * cStar_t                                        (:175)
//   cHat_t     = tanh(att2_fc2(drop(relu(att2_fc1(attended_t)))))           (:176)
// It depends only on the three LSTMs' cell states, so it is evaluated for ALL T steps at once: the four Linears
// are [T*B, .] products on the grouped GEMM (relu / dropout / tanh in its epilogue), and the row-wise glue
// between them lives here: gathering cStar from the three padded cell-state buffers, softmax * cStar, its
// backward, and the scatter of d cStar back onto the cell states (the LSTM kernels' dc_ext input).
#include "internal.h"

namespace mfm {

namespace {

struct CsSrc { const float* cs[3]; float* dcx[3]; int h[3], Hp[3], off[3]; int tot, T, B; };

// cstar[t][b][0:tot] = c_{t-1}, cstar[t][b][tot:2tot] = c_t  (c_{-1} = 0)
__global__ __launch_bounds__(256) void mfn_cstar_kernel(const CsSrc S, float* __restrict__ cstar) {
  const int64_t TB = (int64_t)S.T * S.B;
  const int A2 = 2 * S.tot;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < TB * A2; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / A2;
    const int c = (int)(i - row * A2);
    const int half = c >= S.tot;
    const int cc = half ? c - S.tot : c;
    const int m = (cc >= S.off[1]) + (cc >= S.off[2]);
    const int j = cc - S.off[m];
    const int64_t t = row / S.B, b = row - t * S.B;
    const int64_t ts = half ? t : t - 1;
    cstar[i] = (ts >= 0) ? S.cs[m][(ts * S.B + b) * S.Hp[m] + j] : 0.0f;
  }
}

// dcx_m[t][b][j] = dcs[t][b][tot + off_m + j] + dcs[t+1][b][off_m + j]   (pad units j >= h: 0)
__global__ __launch_bounds__(256) void mfn_dcs_scatter_kernel(const CsSrc S, const float* __restrict__ dcs) {
  const int64_t TB = (int64_t)S.T * S.B;
  const int A2 = 2 * S.tot;
  const int HPS = S.Hp[0] + S.Hp[1] + S.Hp[2];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < TB * HPS; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / HPS;
    const int c = (int)(i - row * HPS);
    const int m = (c >= S.Hp[0]) + (c >= S.Hp[0] + S.Hp[1]);
    const int j = c - (m == 0 ? 0 : (m == 1 ? S.Hp[0] : S.Hp[0] + S.Hp[1]));
    const int64_t t = row / S.B;
    float v = 0.0f;
    if (j < S.h[m]) {
      v = dcs[row * A2 + S.tot + S.off[m] + j];
      if (t + 1 < S.T) v += dcs[(row + S.B) * A2 + S.off[m] + j];
    }
    S.dcx[m][row * S.Hp[m] + j] = v;
  }
}

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

constexpr int SM_MAXPER = 16;      // columns per lane: rows up to 1024 wide

// one wave per row: att <- softmax(att) in place, attended = att * cstar
__global__ __launch_bounds__(256) void mfn_softmax_fwd_kernel(float* __restrict__ att, const float* __restrict__ cstar,
                                                             float* __restrict__ attended, int64_t rows, int n) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* a = att + row * n;
  float v[SM_MAXPER];
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < SM_MAXPER; ++i) {
    const int c = lane + 64 * i;
    v[i] = (c < n) ? a[c] : -3.0e38f;
    mx = fmaxf(mx, v[i]);
  }
  mx = wave_max(mx);
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < SM_MAXPER; ++i) {
    const int c = lane + 64 * i;
    v[i] = (c < n) ? expf(v[i] - mx) : 0.0f;
    s += v[i];
  }
  const float inv = 1.0f / wave_add(s);
#pragma unroll
  for (int i = 0; i < SM_MAXPER; ++i) {
    const int c = lane + 64 * i;
    if (c < n) {
      const float p = v[i] * inv;
      a[c] = p;
      attended[row * n + c] = p * cstar[row * n + c];
    }
  }
}

// d_att = d_attended * cstar ; d_logits = att * (d_att - sum(d_att * att)) ; d_cstar (first part) = d_attended * att
__global__ __launch_bounds__(256) void mfn_softmax_bwd_kernel(const float* __restrict__ datt, const float* __restrict__ att,
                                                             const float* __restrict__ cstar, float* __restrict__ dlog,
                                                             float* __restrict__ dcs, int64_t rows, int n) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float p[SM_MAXPER], g[SM_MAXPER];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < SM_MAXPER; ++i) {
    const int c = lane + 64 * i;
    const bool ok = c < n;
    const float da = ok ? datt[row * n + c] : 0.0f;
    p[i] = ok ? att[row * n + c] : 0.0f;
    const float cs = ok ? cstar[row * n + c] : 0.0f;
    g[i] = da * cs;
    s += g[i] * p[i];
    if (ok) dcs[row * n + c] = da * p[i];
  }
  s = wave_add(s);
#pragma unroll
  for (int i = 0; i < SM_MAXPER; ++i) {
    const int c = lane + 64 * i;
    if (c < n) dlog[row * n + c] = p[i] * (g[i] - s);
  }
}

int grid_for(int64_t n) {
  int64_t nb = (n + 255) / 256;
  if (nb < 1) nb = 1;
  if (nb > 4096) nb = 4096;
  return (int)nb;
}

}  // namespace

int mfn_cstar_launch(const MfnCs& c, float* cstar, hipStream_t stream) {
  CsSrc S;
  memset(&S, 0, sizeof(S));
  int off = 0;
  for (int m = 0; m < 3; ++m) { S.cs[m] = c.cs[m]; S.h[m] = c.h[m]; S.Hp[m] = round_up(c.h[m], 16); S.off[m] = off; off += c.h[m]; }
  S.tot = off; S.T = c.T; S.B = c.B;
  MFM_LAUNCH_TIMED(mfn_cstar_kernel, dim3(grid_for((int64_t)c.T * c.B * 2 * off)), dim3(256), 0, stream, S, cstar);
  MFM_LAUNCH_CHECK("mfn_cstar_kernel");
  return MFM_OK;
}

int mfn_dcs_scatter_launch(const MfnCs& c, const float* dcs, hipStream_t stream) {
  CsSrc S;
  memset(&S, 0, sizeof(S));
  int off = 0, hps = 0;
  for (int m = 0; m < 3; ++m) {
    S.dcx[m] = c.dcx[m]; S.h[m] = c.h[m]; S.Hp[m] = round_up(c.h[m], 16); S.off[m] = off; off += c.h[m]; hps += S.Hp[m];
  }
  S.tot = off; S.T = c.T; S.B = c.B;
  MFM_LAUNCH_TIMED(mfn_dcs_scatter_kernel, dim3(grid_for((int64_t)c.T * c.B * hps)), dim3(256), 0, stream, S, dcs);
  MFM_LAUNCH_CHECK("mfn_dcs_scatter_kernel");
  return MFM_OK;
}

int mfn_softmax_fwd_launch(float* att, const float* cstar, float* attended, int64_t rows, int n, hipStream_t stream) {
  MFM_REQUIRE(n >= 1 && n <= 64 * SM_MAXPER, "mfn softmax: row width %d > %d", n, 64 * SM_MAXPER);
  MFM_LAUNCH_TIMED(mfn_softmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, att, cstar, attended, rows, n);
  MFM_LAUNCH_CHECK("mfn_softmax_fwd_kernel");
  return MFM_OK;
}

int mfn_softmax_bwd_launch(const float* datt, const float* att, const float* cstar, float* dlog, float* dcs, int64_t rows,
                           int n, hipStream_t stream) {
  MFM_REQUIRE(n >= 1 && n <= 64 * SM_MAXPER, "mfn softmax: row width %d > %d", n, 64 * SM_MAXPER);
  MFM_LAUNCH_TIMED(mfn_softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, datt, att, cstar, dlog,
                     dcs, rows, n);
  MFM_LAUNCH_CHECK("mfn_softmax_bwd_kernel");
  return MFM_OK;
}

}  // namespace mfm
