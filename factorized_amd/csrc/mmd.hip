// Maximum-mean-discrepancy regulariser of the non-KL MFM (reference mfm_model.py:14-34) as one kernel:
//   K(a, b) = exp(-mean_d((a - b)^2) / dim)             (the reference divides by dim twice: mean, then /dim)
//   mmd     = mean K(g, g) + mean K(z, z) - 2 mean K(g, z)          g ~ N(0, 1), same shape as z
// and its gradient wrt z in the same pass (the Gaussian sample is a constant):
//   d mmd / d z_i = 4 / (B^2 dim^2) * sum_j [ K(g_j, z_i) (z_i - g_j) - K(z_i, z_j) (z_i - z_j) ]
// The reference materialises three [B, B, dim] tensors per term and four terms per step; here a
// workgroup owns 32 (small batches: 8) rows i, walks j in tiles of 32 staged in LDS (rows padded to an odd stride), computes
// the three kernel values of a pair once, and contracts them against the tile for the gradient.
#include <string.h>

#include <algorithm>

#include "internal.h"

namespace mfm {

namespace {

constexpr int JT = 32;          // columns j per tile
constexpr int MMD_KMAX = 256;   // feature dimension limit (registers: dim / slices accumulators per thread)

// up to 4 independent terms per launch (blockIdx.y): the non-KL MFM regularises z_l, z_a, z_v, z_y in one go
struct MmdTerm { const float* z; const float* g; float* dz; int dim, pad_; };
struct MmdGroup { MmdTerm t[4]; };

// MI rows i per workgroup; the 256 threads form MI rows x NS = 256 / MI slices (of the j tile in phase 1, of the features
// in phase 2).  MI = 32 is the throughput shape; MI = 8 serves small batches, where one 32-row workgroup per term is a
// 43 us latency chain on four CUs (B = 32: 4 workgroups -> 16, each with a quarter of the pairs and of the contraction).
template <int MI>
__global__ __launch_bounds__(256) void mmd_kernel(const MmdGroup G, int B, float* __restrict__ loss, float loss_scale,
                                                  int64_t ldz, int64_t ldg, int64_t lddz, float dz_scale) {
  constexpr int NS = 256 / MI, PJ = JT / NS, NA = MMD_KMAX / NS;
  static_assert(PJ >= 1 && NS * PJ == JT, "slices must tile the j tile");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const MmdTerm& M = G.t[blockIdx.y];
  const float* __restrict__ z = M.z;
  const float* __restrict__ g = M.g;
  float* __restrict__ dz = M.dz;
  const int dim = M.dim;
  const int ld = dim | 1;                         // odd row stride: conflict-free when lanes walk rows
  float* zi = lds;                                // [MI][ld]
  float* gi = zi + MI * ld;
  float* zj = gi + MI * ld;                       // [JT][ld]
  float* gj = zj + JT * ld;
  float* czz = gj + JT * ld;                      // [MI][JT+1]
  float* cgz = czz + MI * (JT + 1);
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int il = tid & (MI - 1), js = tid / MI;   // row of the tile, j-slice / k-slice 0..NS-1
  const int i0 = blockIdx.x * MI;
  const float inv_k2 = 1.0f / ((float)dim * (float)dim);

  for (int e = tid; e < MI * dim; e += 256) {
    const int r = e / dim, c = e - r * dim;
    const int row = min(i0 + r, B - 1);
    zi[r * ld + c] = z[(int64_t)row * ldz + c];
    gi[r * ld + c] = g[(int64_t)row * ldg + c];
  }
  float acc[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) acc[a] = 0.0f;
  float part = 0.0f;
  const bool ilive = i0 + il < B;

  for (int j0 = 0; j0 < B; j0 += JT) {
    __syncthreads();                              // previous tile fully consumed (and zi/gi visible)
    for (int e = tid; e < JT * dim; e += 256) {
      const int r = e / dim, c = e - r * dim;
      const int row = min(j0 + r, B - 1);
      zj[r * ld + c] = z[(int64_t)row * ldz + c];
      gj[r * ld + c] = g[(int64_t)row * ldg + c];
    }
    __syncthreads();
    // phase 1: the three kernel values of (i, j) for this thread's PJ columns
#pragma unroll
    for (int jj = 0; jj < PJ; ++jj) {
      const int jl = js * PJ + jj;
      const float* zr = zi + il * ld; const float* gr = gi + il * ld;
      const float* zc = zj + jl * ld; const float* gc = gj + jl * ld;
      float dzz = 0.0f, dgz = 0.0f, dgg = 0.0f;
      for (int c = 0; c < dim; ++c) {
        const float a = zr[c], b = zc[c], p = gr[c], q = gc[c];
        dzz = fmaf(a - b, a - b, dzz);
        dgz = fmaf(a - q, a - q, dgz);
        dgg = fmaf(p - q, p - q, dgg);
      }
      const bool live = ilive && (j0 + jl < B);
      const float kzz = live ? __expf(-dzz * inv_k2) : 0.0f;
      const float kgz = live ? __expf(-dgz * inv_k2) : 0.0f;
      const float kgg = live ? __expf(-dgg * inv_k2) : 0.0f;
      part += kzz + kgg - 2.0f * kgz;
      czz[il * (JT + 1) + jl] = kzz;
      cgz[il * (JT + 1) + jl] = kgz;
    }
    __syncthreads();
    // phase 2: gradient contraction, this thread owns features c = js + NS a of row il
    if (dz) {
      for (int jl = 0; jl < JT; ++jl) {
        const float a = czz[il * (JT + 1) + jl], b = cgz[il * (JT + 1) + jl];
        const float* zc = zj + jl * ld; const float* gc = gj + jl * ld;
        const float* zr = zi + il * ld;
#pragma unroll
        for (int t = 0; t < NA; ++t) {
          const int c = js + NS * t;
          if (c < dim) {
            const float v = zr[c];
            acc[t] += b * (v - gc[c]) - a * (v - zc[c]);
          }
        }
      }
    }
  }
  if (dz && ilive) {
    const float coef = 4.0f * loss_scale * inv_k2 * dz_scale;     // loss_scale = 1 / B^2
#pragma unroll
    for (int t = 0; t < NA; ++t) {
      const int c = js + NS * t;
      if (c < dim) dz[(int64_t)(i0 + il) * lddz + c] = coef * acc[t];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if ((tid & 63) == 0) red[tid >> 6] = part;
  __syncthreads();
  if (tid == 0 && loss) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * loss_scale);
}


// ---------------------------------------------------------------------------------------------------------------------
// GEMM form for LARGE batches (round 3).  The kernel above walks all B^2 pairs on the VALU with B / 32 workgroups per term:
// 1.1 ms at B = 1024 (44 % of the MFM step).  With |a - b|^2 = |a|^2 + |b|^2 - 2 a.b the three distance matrices are Gram
// matrices and the gradient is two more products -- all on the grouped fp32 MFMA GEMM:
//   1. ZZ = Z Z^T, GZ = Z G^T, GG = G G^T                                   (12 problems, K = dim)
//   2. mmd_k_kernel: K = exp(-(n_i + n_j - 2 gram) / dim^2) in place (the norms are the Gram diagonals), the loss, the row
//      sums s_i = sum_j Kgz_ij - Kzz_ij, and dz_i <- coef s_i z_i
//   3. dz += coef Kzz Z - coef Kgz G                                        (8 accumulating problems, K = B)
// from d mmd / d z_i = 4 / (B^2 dim^2) sum_j [ Kgz_ij (z_i - g_j) - Kzz_ij (z_i - z_j) ].  Scratch: 3 B x ldb floats per term.
struct MmdKTerm { const float* z; float* dz; float* zz; float* gz; const float* gg; int dim, pad_; };
struct MmdKGroup { MmdKTerm t[4]; };

__global__ __launch_bounds__(256) void mmd_k_kernel(const MmdKGroup G, int B, int64_t ldb, int64_t ldz, int64_t lddz,
                                                    float* __restrict__ loss, float loss_scale, float coef_scale) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const MmdKTerm& M = G.t[blockIdx.y];
  float* nz = lds;                 // [B] |z_j|^2
  float* ng = lds + B;             // [B] |g_j|^2
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float inv_k2 = 1.0f / ((float)M.dim * (float)M.dim);
  for (int j = tid; j < B; j += 256) { nz[j] = M.zz[(int64_t)j * ldb + j]; ng[j] = M.gg[(int64_t)j * ldb + j]; }
  __syncthreads();
  float part = 0.0f;
  // one wave per row i: j runs over the lanes (coalesced rows of the three matrices)
  for (int i = blockIdx.x * 4 + wave; i < B; i += gridDim.x * 4) {
    float* zzr = M.zz + (int64_t)i * ldb;
    float* gzr = M.gz + (int64_t)i * ldb;
    const float* ggr = M.gg + (int64_t)i * ldb;
    const float nzi = nz[i], ngi = ng[i];
    float rs = 0.0f;
    for (int j = lane; j < B; j += 64) {
      const float kzz = __expf(-fmaxf(nzi + nz[j] - 2.0f * zzr[j], 0.0f) * inv_k2);
      const float kgz = __expf(-fmaxf(nzi + ng[j] - 2.0f * gzr[j], 0.0f) * inv_k2);
      const float kgg = __expf(-fmaxf(ngi + ng[j] - 2.0f * ggr[j], 0.0f) * inv_k2);
      part += kzz + kgg - 2.0f * kgz;
      rs += kgz - kzz;
      zzr[j] = kzz; gzr[j] = kgz;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rs += __shfl_xor(rs, o, 64);
    if (M.dz) {
      const float c = 4.0f * loss_scale * inv_k2 * coef_scale * rs;
      for (int k = lane; k < M.dim; k += 64) M.dz[(int64_t)i * lddz + k] = c * M.z[(int64_t)i * ldz + k];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if (lane == 0) red[wave] = part;
  __syncthreads();
  if (tid == 0 && loss) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * loss_scale);
}

}  // namespace

int64_t mmd_scratch_floats(int B, int count) { return (int64_t)count * 3 * B * round_up(B, 4); }

static int mmd_gemm_form(const MmdItem* items, int count, int64_t ldz, int64_t ldg, int64_t lddz, int B, float* loss,
                         float dz_scale, float* scratch, hipStream_t stream) {
  const int64_t ldb = round_up(B, 4);
  MFM_REQUIRE((((uintptr_t)scratch) & 15) == 0 && 2 * (size_t)B * sizeof(float) <= 64 * 1024, "mmd (gemm form): scratch alignment / B = %d", B);
  MfmGemmDesc g[12];
  memset(g, 0, sizeof(g));
  MmdKGroup K;
  memset(&K, 0, sizeof(K));
  int n = 0;
  for (int e = 0; e < count; ++e) {
    float* zz = scratch + (int64_t)(3 * e) * B * ldb;
    float* gz = zz + (int64_t)B * ldb;
    float* gg = gz + (int64_t)B * ldb;
    const float* A[3] = {items[e].z, items[e].z, items[e].g};
    const int64_t lA[3] = {ldz, ldz, ldg};
    const float* Bm[3] = {items[e].z, items[e].g, items[e].g};
    const int64_t lB[3] = {ldz, ldg, ldg};
    float* C[3] = {zz, gz, gg};
    for (int p = 0; p < 3; ++p) {
      MfmGemmDesc& d = g[n++];
      d.a = A[p]; d.a_sm = lA[p]; d.a_sk = 1;
      d.b = Bm[p]; d.b_sk = 1; d.b_sn = lB[p];              // B^T: element (k, n) = row n, column k
      d.c = C[p]; d.ldc = ldb;
      d.m = B; d.n = B; d.n_valid = B; d.k = items[e].dim; d.batch = 1; d.split_k = 1; d.alpha = 1.0f;
    }
    K.t[e].z = items[e].z; K.t[e].dz = items[e].dz; K.t[e].zz = zz; K.t[e].gz = gz; K.t[e].gg = gg; K.t[e].dim = items[e].dim;
  }
  int rc = gemm_group_launch(g, n, stream);
  if (rc != MFM_OK) return rc;
  const float loss_scale = 1.0f / ((float)B * (float)B);
  const dim3 grid(std::min(cdiv(B, 4), 2 * device_cus()), count);
  MFM_LAUNCH_TIMED(mmd_k_kernel, grid, dim3(256), 2 * (size_t)B * sizeof(float), stream, K, B, ldb, ldz, lddz, loss, loss_scale, dz_scale);
  MFM_LAUNCH_CHECK("mmd_k_kernel");
  bool any_dz = false;
  for (int e = 0; e < count; ++e) any_dz = any_dz || items[e].dz;
  if (!any_dz) return MFM_OK;
  memset(g, 0, sizeof(g));
  n = 0;
  for (int e = 0; e < count; ++e) {
    if (!items[e].dz) continue;
    const float coef = 4.0f * loss_scale * dz_scale / ((float)items[e].dim * (float)items[e].dim);
    for (int p = 0; p < 2; ++p) {
      MfmGemmDesc& d = g[n++];
      d.a = p == 0 ? K.t[e].zz : K.t[e].gz; d.a_sm = ldb; d.a_sk = 1;
      d.b = p == 0 ? items[e].z : items[e].g; d.b_sk = p == 0 ? ldz : ldg; d.b_sn = 1;
      d.c = items[e].dz; d.ldc = lddz;
      d.m = B; d.n = items[e].dim; d.n_valid = items[e].dim; d.k = B; d.batch = 1; d.split_k = 0; d.accumulate = 1;
      d.alpha = p == 0 ? coef : -coef;
    }
  }
  return gemm_group_launch(g, n, stream);
}

int mmd_group_launch(const MmdItem* items, int count, int64_t ldz, int64_t ldg, int64_t lddz, int B, float* loss,
                     float dz_scale, hipStream_t stream, float* scratch) {
  MFM_REQUIRE(B >= 1 && count >= 1 && count <= 4, "mmd: B=%d terms=%d", B, count);
  // large batches with scratch from the caller: the GEMM form (measured crossover: plan.hip)
  if (scratch) {
    bool ok = B <= 8192;
    for (int i = 0; i < count; ++i)
      ok = ok && items[i].z && items[i].g && items[i].dim >= 1 && (ldz & 3) == 0 && (ldg & 3) == 0 &&
           (((uintptr_t)items[i].z) & 15) == 0 && (((uintptr_t)items[i].g) & 15) == 0;
    if (ok) return mmd_gemm_form(items, count, ldz, ldg, lddz, B, loss, dz_scale, scratch, stream);
  }
  MmdGroup G;
  memset(&G, 0, sizeof(G));
  int dmax = 0;
  for (int i = 0; i < count; ++i) {
    MFM_REQUIRE(items[i].dim >= 1 && items[i].z && items[i].g, "mmd: term %d", i);
    if (items[i].dim > MMD_KMAX) { set_error("mmd: feature dimension %d > %d", items[i].dim, MMD_KMAX); return MFM_ERR_UNSUPPORTED; }
    G.t[i].z = items[i].z; G.t[i].g = items[i].g; G.t[i].dz = items[i].dz; G.t[i].dim = items[i].dim;
    dmax = std::max(dmax, items[i].dim);
  }
  const int ld = dmax | 1;
  // small batches: 8-row workgroups (4x the workgroups, a quarter of the serial work each); MFM_MMD_ROWS=8|32 overrides
  int MI = (B <= 256) ? 8 : 32;
  if (const char* e = opt_get("MFM_MMD_ROWS")) MI = (atoi(e) == 8) ? 8 : 32;
  const size_t lds = ((size_t)(2 * MI + 2 * JT) * ld + 2 * (size_t)MI * (JT + 1)) * sizeof(float);
  const void* fn = (MI == 8) ? (const void*)mmd_kernel<8> : (const void*)mmd_kernel<32>;
  if (lds > 64 * 1024) MFM_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const dim3 grid(cdiv(B, MI), count);
  if (MI == 8)
    MFM_LAUNCH_TIMED(mmd_kernel<8>, grid, dim3(256), lds, stream, G, B, loss, 1.0f / ((float)B * (float)B), ldz, ldg, lddz, dz_scale);
  else
    MFM_LAUNCH_TIMED(mmd_kernel<32>, grid, dim3(256), lds, stream, G, B, loss, 1.0f / ((float)B * (float)B), ldz, ldg, lddz, dz_scale);
  MFM_LAUNCH_CHECK("mmd_kernel");
  return MFM_OK;
}

int mmd_launch(const float* z, int64_t ldz, const float* g, int64_t ldg, int B, int dim, float* loss, float* dz, int64_t lddz,
               float dz_scale, hipStream_t stream) {
  MmdItem it = {z, g, dz, dim};
  return mmd_group_launch(&it, 1, ldz, ldg, lddz, B, loss, dz_scale, stream, nullptr);
}

}  // namespace mfm

using namespace mfm;

extern "C" int mfm_mmd_fwd_bwd(const float* z, const float* gauss, int32_t B, int32_t dim, float* loss, float* dz,
                               void* stream) {
  if (!z || !gauss || !loss) { set_error("mfm_mmd_fwd_bwd: null argument"); return MFM_ERR_ARG; }
  return mmd_launch(z, dim, gauss, dim, (int)B, (int)dim, loss, dz, dim, 1.0f, (hipStream_t)stream);
}
