// Maximum-mean-discrepancy regulariser of the non-KL MFM (reference mfm_model.py:14-34) as one kernel:
//   K(a, b) = exp(-mean_d((a - b)^2) / dim)             (the reference divides by dim twice: mean, then /dim)
//   mmd     = mean K(g, g) + mean K(z, z) - 2 mean K(g, z)          g ~ N(0, 1), same shape as z
// and its gradient wrt z in the same pass (the Gaussian sample is a constant):
//   d mmd / d z_i = 4 / (B^2 dim^2) * sum_j [ K(g_j, z_i) (z_i - g_j) - K(z_i, z_j) (z_i - z_j) ]
// The reference materialises three [B, B, dim] tensors per term and four terms per step; here a
// workgroup owns 32 (small batches: 8) rows i, walks j in tiles of 32 staged in LDS (rows padded to an odd stride), computes
// the three kernel values of a pair once, and contracts them against the tile for the gradient.
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

}  // namespace

int mmd_group_launch(const MmdItem* items, int count, int64_t ldz, int64_t ldg, int64_t lddz, int B, float* loss,
                     float dz_scale, hipStream_t stream) {
  MFM_REQUIRE(B >= 1 && count >= 1 && count <= 4, "mmd: B=%d terms=%d", B, count);
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
  if (const char* e = getenv("MFM_MMD_ROWS")) MI = (atoi(e) == 8) ? 8 : 32;
  const size_t lds = ((size_t)(2 * MI + 2 * JT) * ld + 2 * (size_t)MI * (JT + 1)) * sizeof(float);
  const void* fn = (MI == 8) ? (const void*)mmd_kernel<8> : (const void*)mmd_kernel<32>;
  if (lds > 64 * 1024) MFM_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const dim3 grid(cdiv(B, MI), count);
  if (MI == 8)
    hipLaunchKernelGGL(mmd_kernel<8>, grid, dim3(256), lds, stream, G, B, loss, 1.0f / ((float)B * (float)B), ldz, ldg, lddz, dz_scale);
  else
    hipLaunchKernelGGL(mmd_kernel<32>, grid, dim3(256), lds, stream, G, B, loss, 1.0f / ((float)B * (float)B), ldz, ldg, lddz, dz_scale);
  MFM_LAUNCH_CHECK("mmd_kernel");
  return MFM_OK;
}

int mmd_launch(const float* z, int64_t ldz, const float* g, int64_t ldg, int B, int dim, float* loss, float* dz, int64_t lddz,
               float dz_scale, hipStream_t stream) {
  MmdItem it = {z, g, dz, dim};
  return mmd_group_launch(&it, 1, ldz, ldg, lddz, B, loss, dz_scale, stream);
}

}  // namespace mfm

using namespace mfm;

extern "C" int mfm_mmd_fwd_bwd(const float* z, const float* gauss, int32_t B, int32_t dim, float* loss, float* dz,
                               void* stream) {
  if (!z || !gauss || !loss) { set_error("mfm_mmd_fwd_bwd: null argument"); return MFM_ERR_ARG; }
  return mmd_launch(z, dim, gauss, dim, (int)B, (int)dim, loss, dz, dim, 1.0f, (hipStream_t)stream);
}
