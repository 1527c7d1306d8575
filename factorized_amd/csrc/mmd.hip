// Maximum-mean-discrepancy regulariser of the non-KL MFM (reference mfm_model.py:14-34) as one kernel:
//   K(a, b) = exp(-mean_d((a - b)^2) / dim)             (the reference divides by dim twice: mean, then /dim)
//   mmd     = mean K(g, g) + mean K(z, z) - 2 mean K(g, z)          g ~ N(0, 1), same shape as z
// and its gradient wrt z in the same pass (the Gaussian sample is a constant):
//   d mmd / d z_i = 4 / (B^2 dim^2) * sum_j [ K(g_j, z_i) (z_i - g_j) - K(z_i, z_j) (z_i - z_j) ]
// The reference materialises three [B, B, dim] tensors per term and four terms per step; here a
// workgroup owns 32 rows i, walks j in tiles of 32 staged in LDS (rows padded to an odd stride), computes
// the three kernel values of a pair once, and contracts them against the tile for the gradient.
#include "internal.h"

namespace mfm {

namespace {

constexpr int MT = 32;          // rows per tile
constexpr int MMD_KMAX = 256;   // feature dimension limit (registers: dim/8 accumulators per thread)

// up to 4 independent terms per launch (blockIdx.y): the non-KL MFM regularises z_l, z_a, z_v, z_y in one go
struct MmdTerm { const float* z; const float* g; float* dz; int dim, pad_; };
struct MmdGroup { MmdTerm t[4]; };

__global__ __launch_bounds__(256) void mmd_kernel(const MmdGroup G, int B, float* __restrict__ loss, float loss_scale,
                                                  int64_t ldz, int64_t ldg, int64_t lddz, float dz_scale) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const MmdTerm& M = G.t[blockIdx.y];
  const float* __restrict__ z = M.z;
  const float* __restrict__ g = M.g;
  float* __restrict__ dz = M.dz;
  const int dim = M.dim;
  const int ld = dim | 1;                         // odd row stride: conflict-free when lanes walk rows
  float* zi = lds;                                // [MT][ld]
  float* gi = zi + MT * ld;
  float* zj = gi + MT * ld;
  float* gj = zj + MT * ld;
  float* czz = gj + MT * ld;                      // [MT][MT+1]
  float* cgz = czz + MT * (MT + 1);
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int il = tid & (MT - 1), js = tid >> 5;   // row of the tile, j-slice / k-slice 0..7
  const int i0 = blockIdx.x * MT;
  const float inv_k2 = 1.0f / ((float)dim * (float)dim);

  for (int e = tid; e < MT * dim; e += 256) {
    const int r = e / dim, c = e - r * dim;
    const int row = min(i0 + r, B - 1);
    zi[r * ld + c] = z[(int64_t)row * ldz + c];
    gi[r * ld + c] = g[(int64_t)row * ldg + c];
  }
  float acc[MMD_KMAX / 8];
#pragma unroll
  for (int a = 0; a < MMD_KMAX / 8; ++a) acc[a] = 0.0f;
  float part = 0.0f;
  const bool ilive = i0 + il < B;

  for (int j0 = 0; j0 < B; j0 += MT) {
    __syncthreads();                              // previous tile fully consumed (and zi/gi visible)
    for (int e = tid; e < MT * dim; e += 256) {
      const int r = e / dim, c = e - r * dim;
      const int row = min(j0 + r, B - 1);
      zj[r * ld + c] = z[(int64_t)row * ldz + c];
      gj[r * ld + c] = g[(int64_t)row * ldg + c];
    }
    __syncthreads();
    // phase 1: the three kernel values of (i, j) for this thread's 4 columns
#pragma unroll
    for (int jj = 0; jj < MT / 8; ++jj) {
      const int jl = js * (MT / 8) + jj;
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
      czz[il * (MT + 1) + jl] = kzz;
      cgz[il * (MT + 1) + jl] = kgz;
    }
    __syncthreads();
    // phase 2: gradient contraction, this thread owns features c = js + 8 a of row il
    if (dz) {
      for (int jl = 0; jl < MT; ++jl) {
        const float a = czz[il * (MT + 1) + jl], b = cgz[il * (MT + 1) + jl];
        const float* zc = zj + jl * ld; const float* gc = gj + jl * ld;
        const float* zr = zi + il * ld;
#pragma unroll
        for (int t = 0; t < MMD_KMAX / 8; ++t) {
          const int c = js + 8 * t;
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
    for (int t = 0; t < MMD_KMAX / 8; ++t) {
      const int c = js + 8 * t;
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
  const size_t lds = ((size_t)4 * MT * ld + 2 * MT * (MT + 1)) * sizeof(float);
  if (lds > 64 * 1024)
    MFM_HIP_CHECK(hipFuncSetAttribute((const void*)mmd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(mmd_kernel, dim3(cdiv(B, MT), count), dim3(256), lds, stream, G, B, loss, 1.0f / ((float)B * (float)B),
                     ldz, ldg, lddz, dz_scale);
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
