// HBM-bound helpers of the MFM step: reconstruction loss (+ its gradient), fused flat Adam, fill.
#include <math.h>
#include <stdarg.h>

#include "internal.h"
#include "lstamp.h"

namespace mfm {

// ---------------------------------------------------------------- error plumbing (host)
static thread_local char g_err[512] = "";
thread_local LaunchEvents* tls_launch_events = nullptr;

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int hip_fail(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
  return MFM_ERR_HIP;
}

// ---------------------------------------------------------------- option tables (common.h)
}  // namespace mfm
#include <stdlib.h>
#include <string>
#include <unordered_map>
extern char** environ;
namespace mfm {
struct OptTable { std::unordered_map<std::string, std::string> kv; };
static thread_local const OptTable* g_opt_scope = nullptr;
const char* opt_get(const char* name) {
  if (g_opt_scope) {
    auto it = g_opt_scope->kv.find(name);
    return it == g_opt_scope->kv.end() ? nullptr : it->second.c_str();
  }
  return ::getenv(name);
}
OptTable* opt_table_from_env() {
  OptTable* t = new OptTable();
  for (char** e = environ; e && *e; ++e) {
    if (strncmp(*e, "MFM_", 4) != 0) continue;
    const char* eq = strchr(*e, '=');
    if (!eq) continue;
    t->kv[std::string(*e, eq - *e)] = std::string(eq + 1);
  }
  return t;
}
void opt_table_set(OptTable* t, const char* name, const char* value) {
  if (!t || !name) return;
  if (value) t->kv[name] = value; else t->kv.erase(name);
}
void opt_table_free(OptTable* t) { delete t; }
OptScope::OptScope(const OptTable* t) : prev(g_opt_scope) { g_opt_scope = t; }
OptScope::~OptScope() { g_opt_scope = prev; }

// ---------------------------------------------------------------- block reduction
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over the block; result valid in thread 0.  `scratch` >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  float r = 0.0f;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += scratch[i];
  }
  __syncthreads();
  return r;
}

// ---------------------------------------------------------------- MSE fwd+bwd (grouped)
struct MseGroup { MseItem it[3]; int count; };

__global__ __launch_bounds__(256) void mse_kernel(const MseGroup g) {
  __shared__ float scratch[16];
  int gi = 0;
#pragma unroll 1
  for (int i = 1; i < g.count; ++i)
    if ((int)blockIdx.x >= g.it[i].block_begin) gi = i;
  const MseItem& it = g.it[gi];
  const int nblk = ((gi + 1 < g.count) ? g.it[gi + 1].block_begin : (int)gridDim.x) - it.block_begin;
  const int64_t total = it.rows * it.d;
  float part = 0.0f;
  for (int64_t i = (int64_t)(blockIdx.x - it.block_begin) * 256 + threadIdx.x; i < total; i += (int64_t)nblk * 256) {
    const int64_t r = i / it.d;
    const int c = (int)(i - r * it.d);
    const float diff = it.xhat[i] - it.x[r * it.ldx + c];
    part += diff * diff;
    if (it.dxhat) it.dxhat[i] = it.grad_scale * diff;
  }
  const float s = block_sum(part, scratch);
  if (threadIdx.x == 0 && it.loss_slot) atomicAdd(it.loss_slot, s * it.inv_count);
}

int mse_group_launch(const MseItem* items, int count, hipStream_t stream) {
  MseGroup g;
  memset(&g, 0, sizeof(g));
  g.count = count;
  int total = 0;
  for (int i = 0; i < count; ++i) {
    g.it[i] = items[i];
    g.it[i].block_begin = total;
    int64_t n = items[i].rows * items[i].d;
    int nb = (int)((n + 1023) / 1024);
    if (nb < 1) nb = 1;
    if (nb > 1024) nb = 1024;
    total += nb;
  }
  MFM_LAUNCH_TIMED(mse_kernel, dim3(total), dim3(256), 0, stream, g);
  MFM_LAUNCH_CHECK("mse_kernel");
  return MFM_OK;
}

// ---------------------------------------------------------------- Adam (flat)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   float beta1, float beta2, float eps, float step_size,
                                                   float bc2_sqrt, float grad_scale, const float* __restrict__ guard) {
  // guard word (mfm_adam_flat_guarded): anything but 0.0f -- the plan stores a NaN -- means the gradients of this step cannot
  // be trusted; p, m and v stay as they are (uniform branch, one cached load per thread)
  LSTAMP(5, 0);
  if (guard && !(guard[0] == 0.0f)) return;
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gg = gv[j] * grad_scale;
      mv[j] = mv[j] + (1.0f - beta1) * (gg - mv[j]);
      vv[j] = vv[j] * beta2 + (1.0f - beta2) * gg * gg;
      const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
      pv[j] = pv[j] - step_size * mv[j] / denom;
    }
    reinterpret_cast<f32x4*>(p)[i] = pv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gg = g[i] * grad_scale;
    const float mm = m[i] + (1.0f - beta1) * (gg - m[i]);
    const float vv = v[i] * beta2 + (1.0f - beta2) * gg * gg;
    m[i] = mm; v[i] = vv;
    p[i] = p[i] - step_size * mm / (sqrtf(vv) / bc2_sqrt + eps);
  }
  LSTAMP_W(5, 15);
}

int adam_launch(float* p, const float* g, float* m, float* v, int64_t n, int step, float lr, float beta1,
                float beta2, float eps, float grad_scale, hipStream_t stream, const float* guard) {
  MFM_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam: bad arguments (n=%lld step=%d)", (long long)n, step);
  MFM_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  int64_t nb = ((n >> 2) + 255) / 256;
  if (nb < 1) nb = 1;
  if (nb > 2048) nb = 2048;
  LSTAMP_BIND();
  MFM_LAUNCH_TIMED(adam_kernel, dim3((int)nb), dim3(256), 0, stream, p, g, m, v, n, beta1, beta2, eps, step_size,
                     bc2_sqrt, grad_scale, guard);
  MFM_LAUNCH_CHECK("adam_kernel");
  return MFM_OK;
}

// ---------------------------------------------------------------- Adam with its step count and learning rate in device memory
// A captured hipGraph freezes kernel arguments: the bias corrections of adam_kernel (functions of the step count, formed on the
// host) would be those of the capture call on every replay.  Here thread 0 of each workgroup forms them from a device counter
// (double precision, like the host path), and a one-thread launch behind it advances the counter -- unless the guard says the
// step was skipped.
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                       float beta1, float beta2, float eps, float grad_scale,
                                                       const int* __restrict__ step_dev, const float* __restrict__ lr_dev,
                                                       const float* __restrict__ guard) {
  if (guard && !(guard[0] == 0.0f)) return;
  __shared__ float sh[2];
  if (threadIdx.x == 0) {
    const double step = (double)(*step_dev + 1);
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    sh[0] = (float)((double)*lr_dev / bc1);
    sh[1] = (float)sqrt(bc2);
  }
  __syncthreads();
  const float step_size = sh[0], bc2_sqrt = sh[1];
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gg = gv[j] * grad_scale;
      mv[j] = mv[j] + (1.0f - beta1) * (gg - mv[j]);
      vv[j] = vv[j] * beta2 + (1.0f - beta2) * gg * gg;
      const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
      pv[j] = pv[j] - step_size * mv[j] / denom;
    }
    reinterpret_cast<f32x4*>(p)[i] = pv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
}
__global__ void adam_step_tick_kernel(int* step_dev, const float* guard) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && !(guard && !(guard[0] == 0.0f))) *step_dev += 1;
}

// ---------------------------------------------------------------- Adam over spans of the flat buffer
// Staged training (train_beta_vae, reference mfm_mosi.py:278-281) leaves whole groups of tensors without a
// gradient; torch.optim.Adam skips a parameter whose .grad is None and keeps a step counter PER PARAMETER, so a
// group that joins later starts its bias correction at step 1.  One launch updates up to MFM_ADAM_MAX_SPANS
// disjoint element ranges, each with its own step count; elements outside every span are left untouched.
struct AdamSpansDev {
  int64_t b4[MFM_ADAM_MAX_SPANS], e4[MFM_ADAM_MAX_SPANS];       // [begin, end) in float4 units
  float step_size[MFM_ADAM_MAX_SPANS], bc2_sqrt[MFM_ADAM_MAX_SPANS];
  int count;
};
__global__ __launch_bounds__(256) void adam_spans_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, int64_t n4,
                                                         const AdamSpansDev S, float beta1, float beta2, float eps,
                                                         float grad_scale, const float* __restrict__ guard) {
  if (guard && !(guard[0] == 0.0f)) return;          // (adam_kernel)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float step_size = 0.0f, bc2_sqrt = 1.0f;
    bool live = false;
#pragma unroll
    for (int k = 0; k < MFM_ADAM_MAX_SPANS; ++k) {
      const bool in = (k < S.count) && (i >= S.b4[k]) && (i < S.e4[k]);
      step_size = in ? S.step_size[k] : step_size;
      bc2_sqrt = in ? S.bc2_sqrt[k] : bc2_sqrt;
      live = live || in;
    }
    if (!live) continue;
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gg = gv[j] * grad_scale;
      mv[j] = mv[j] + (1.0f - beta1) * (gg - mv[j]);
      vv[j] = vv[j] * beta2 + (1.0f - beta2) * gg * gg;
      const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
      pv[j] = pv[j] - step_size * mv[j] / denom;
    }
    reinterpret_cast<f32x4*>(p)[i] = pv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
}

int adam_spans_launch(float* p, const float* g, float* m, float* v, const MfmAdamSpan* spans, int nspans, float lr,
                      float beta1, float beta2, float eps, float grad_scale, hipStream_t stream, const float* guard) {
  MFM_REQUIRE(p && g && m && v && spans && nspans >= 1 && nspans <= MFM_ADAM_MAX_SPANS, "adam spans: bad arguments (nspans=%d)", nspans);
  MFM_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam spans: buffers must be 16-byte aligned");
  AdamSpansDev S;
  memset(&S, 0, sizeof(S));
  S.count = nspans;
  int64_t hi = 0;
  for (int k = 0; k < nspans; ++k) {
    const MfmAdamSpan& sp = spans[k];
    MFM_REQUIRE(sp.begin >= 0 && sp.end > sp.begin && (sp.begin & 3) == 0 && (sp.end & 3) == 0 && sp.step >= 1,
                "adam spans[%d]: [%lld,%lld) step %d (bounds must be multiples of 4 elements, step >= 1)", k,
                (long long)sp.begin, (long long)sp.end, sp.step);
    const double bc1 = 1.0 - pow((double)beta1, (double)sp.step);
    const double bc2 = 1.0 - pow((double)beta2, (double)sp.step);
    S.b4[k] = sp.begin >> 2; S.e4[k] = sp.end >> 2;
    S.step_size[k] = (float)((double)lr / bc1);
    S.bc2_sqrt[k] = (float)sqrt(bc2);
    if (sp.end > hi) hi = sp.end;
  }
  const int64_t n4 = hi >> 2;
  int64_t nb = (n4 + 255) / 256;
  if (nb < 1) nb = 1;
  if (nb > 2048) nb = 2048;
  MFM_LAUNCH_TIMED(adam_spans_kernel, dim3((int)nb), dim3(256), 0, stream, p, g, m, v, n4, S, beta1, beta2, eps, grad_scale, guard);
  MFM_LAUNCH_CHECK("adam_spans_kernel");
  return MFM_OK;
}

// ---------------------------------------------------------------- fill
__global__ void fill_kernel(float* p, int64_t n, float val) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = val;
}
int fill_launch(float* p, int64_t n, float val, hipStream_t stream) {
  int nb = (int)((n + 255) / 256);
  if (nb < 1) nb = 1;
  if (nb > 1024) nb = 1024;
  MFM_LAUNCH_TIMED(fill_kernel, dim3(nb), dim3(256), 0, stream, p, n, val);
  MFM_LAUNCH_CHECK("fill_kernel");
  return MFM_OK;
}

}  // namespace mfm

extern "C" int mfm_abi_version(void) { return MFM_ABI_VERSION; }
extern "C" const char* mfm_last_error(void) { return mfm::g_err; }

extern "C" int mfm_mse_fwd_bwd(const float* xhat, const float* x, int64_t ldx, int64_t rows, int32_t d,
                               float inv_count, float grad_scale, float* dxhat, float* loss_slot, void* stream) {
  if (!xhat || !x || rows <= 0 || d <= 0) {
    mfm::set_error("mfm_mse_fwd_bwd: bad arguments");
    return MFM_ERR_ARG;
  }
  mfm::MseItem it;
  memset(&it, 0, sizeof(it));
  it.xhat = xhat; it.x = x; it.dxhat = dxhat; it.loss_slot = loss_slot;
  it.ldx = ldx; it.rows = rows; it.d = d; it.inv_count = inv_count; it.grad_scale = grad_scale;
  return mfm::mse_group_launch(&it, 1, (hipStream_t)stream);
}

extern "C" int mfm_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, int32_t step, float lr,
                             float beta1, float beta2, float eps, float grad_scale, void* stream) {
  return mfm::adam_launch(p, g, m, v, n, step, lr, beta1, beta2, eps, grad_scale, (hipStream_t)stream);
}

extern "C" int mfm_adam_flat_guarded(float* p, const float* g, float* m, float* v, int64_t n, int32_t step, float lr,
                                     float beta1, float beta2, float eps, float grad_scale, const float* guard, void* stream) {
  return mfm::adam_launch(p, g, m, v, n, step, lr, beta1, beta2, eps, grad_scale, (hipStream_t)stream, guard);
}
extern "C" int mfm_adam_flat_spans_guarded(float* p, const float* g, float* m, float* v, const MfmAdamSpan* spans, int32_t nspans,
                                           float lr, float beta1, float beta2, float eps, float grad_scale, const float* guard,
                                           void* stream) {
  return mfm::adam_spans_launch(p, g, m, v, spans, nspans, lr, beta1, beta2, eps, grad_scale, (hipStream_t)stream, guard);
}

extern "C" int mfm_adam_flat_dev(float* p, const float* g, float* m, float* v, int64_t n, int32_t* step_dev, const float* lr_dev,
                                 float beta1, float beta2, float eps, float grad_scale, const float* guard, void* stream) {
  using namespace mfm;
  MFM_REQUIRE(p && g && m && v && step_dev && lr_dev && n > 0 && (n & 3) == 0, "mfm_adam_flat_dev: bad arguments (n=%lld, a multiple of 4)", (long long)n);
  MFM_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "mfm_adam_flat_dev: buffers must be 16-byte aligned");
  int64_t nb = ((n >> 2) + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipStream_t s = (hipStream_t)stream;
  MFM_LAUNCH_TIMED(adam_dev_kernel, dim3((int)nb), dim3(256), 0, s, p, g, m, v, n, beta1, beta2, eps, grad_scale,
                     (const int*)step_dev, lr_dev, guard);
  MFM_LAUNCH_CHECK("adam_dev_kernel");
  MFM_LAUNCH_TIMED(adam_step_tick_kernel, dim3(1), dim3(64), 0, s, (int*)step_dev, guard);
  MFM_LAUNCH_CHECK("adam_step_tick_kernel");
  return MFM_OK;
}

extern "C" int mfm_adam_flat_spans(float* p, const float* g, float* m, float* v, const MfmAdamSpan* spans, int32_t nspans,
                                   float lr, float beta1, float beta2, float eps, float grad_scale, void* stream) {
  return mfm::adam_spans_launch(p, g, m, v, spans, nspans, lr, beta1, beta2, eps, grad_scale, (hipStream_t)stream);
}
