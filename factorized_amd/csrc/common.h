// Shared device/host helpers for libmfm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "mfm_hip.h"

namespace mfm {

// ---------------------------------------------------------------- host-side error plumbing
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

// ---------------------------------------------------------------- tuning / test switches ("MFM_*")
// Every switch is looked up through opt_get().  Inside a call of a PLAN (mfm_plan_forward / backward / steps) it answers from
// the plan's own table: the MFM_* environment as it was when the plan was CREATED, plus whatever mfm_plan_set_option_str
// changed since -- no getenv on the launch path, and two plans of one process can run with different switches.  Outside a
// plan call (the granular C entry points: GEMM, recurrences, ...) it reads the environment.
const char* opt_get(const char* name);
struct OptTable;                               // name -> value
OptTable* opt_table_from_env();                // snapshot of every MFM_* variable
void opt_table_set(OptTable* t, const char* name, const char* value /* null: remove */);
void opt_table_free(OptTable* t);
struct OptScope {                              // RAII: opt_get() answers from `t` on this thread while the scope lives
  const OptTable* prev;
  explicit OptScope(const OptTable* t);
  ~OptScope();
};

#define MFM_HIP_CHECK(expr)                                   \
  do {                                                        \
    hipError_t _e = (expr);                                   \
    if (_e != hipSuccess) return ::mfm::hip_fail(_e, #expr);  \
  } while (0)

#define MFM_LAUNCH_CHECK(name)                                \
  do {                                                        \
    hipError_t _e = hipGetLastError();                        \
    if (_e != hipSuccess) return ::mfm::hip_fail(_e, name);   \
  } while (0)

#define MFM_REQUIRE(cond, ...)                                \
  do {                                                        \
    if (!(cond)) {                                            \
      ::mfm::set_error(__VA_ARGS__);                          \
      return MFM_ERR_ARG;                                     \
    }                                                         \
  } while (0)

// ---------------------------------------------------------------- kernel time without the bracket (round 6)
// The plan's per-kernel timer used to be two event records around a launch: that reads kernel + ~5 us of packet overhead, and
// subtracting the cost of an empty bracket over-corrects (round 5: the bench line's dominant kernel came out 1.5 us BELOW
// rocprofv3's fastest sample).  A launch issued through MFM_LAUNCH_TIMED while a timer is active hands the timer's two events to
// hipExtLaunchKernelGGL: they then carry the dispatch's own begin / end timestamps -- what rocprofv3's kernel trace reports.
// Only the first launch of a timed region is taken that way; a region of several launches keeps the bracket (Timer,
// plan_internal.h).  No timer active (every step outside a sampled timing call): a plain launch.
struct LaunchEvents { hipEvent_t a, b; int launches; };
extern thread_local LaunchEvents* tls_launch_events;
#define MFM_LAUNCH_TIMED(kernel, grid, block, lds, stream, ...)                                                      \
  do {                                                                                                               \
    ::mfm::LaunchEvents* _le = ::mfm::tls_launch_events;                                                             \
    if (_le && _le->launches++ == 0) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, _le->a, _le->b, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                          \
  } while (0)

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline int64_t round_up64(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- device helpers
typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x4_f32: D = A[16x4] * B[4x16] + C, exact fp32 FMA chain (k ascending).
// lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15];
// result register r of lane l is D[row = (l>>4)*4 + r][col = l&15].
__device__ __forceinline__ f32x4 mma16x16x4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// MFM_FAST_ACT=1 (default): sigmoid/tanh on the hardware transcendentals v_exp_f32 / v_rcp_f32
// (~1 ulp each; absolute error ~1e-7, far inside the 1e-4 parity budget).  MFM_FAST_ACT=0 uses
// the ocml expf/tanhf, which cost ~5x more VALU issue slots per LSTM step.
// Workgroup barrier for LDS hand-offs inside the time loops: only the LDS traffic is drained (lgkmcnt) before s_barrier; the
// "memory" clobber keeps the compiler from moving LDS accesses across it.  Neither this nor `__syncthreads()` waits for
// outstanding GLOBAL stores on gfx950 (a workgroup-scope release needs no vmcnt wait outside threadgroup-split mode: measured
// round 4, profiles/r04_handover_safety.txt) -- whoever announces global data to another workgroup behind a barrier must wait for
// its stores itself: sync_stores() in lstm_seq_small.hip (s_waitcnt vmcnt(0) + barrier).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifndef MFM_FAST_ACT
#define MFM_FAST_ACT 1
#endif

__device__ __forceinline__ float act_sigmoid(float x) {
#if MFM_FAST_ACT
  return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
#else
  return 1.0f / (1.0f + expf(-x));
#endif
}
__device__ __forceinline__ float act_tanh(float x) {
#if MFM_FAST_ACT
  // 2*sigmoid(2x)-1, abs error ~1e-7
  return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f;
#else
  return tanhf(x);
#endif
}

// sc == 1: sigmoid(x); sc == 2: tanh(x).  One instruction stream for lanes that need different gates.
__device__ __forceinline__ float act_scaled(float x, float sc) {
#if MFM_FAST_ACT
  return fmaf(__builtin_amdgcn_rcpf(1.0f + __expf(-sc * x)), sc, 1.0f - sc);
#else
  return sc == 2.0f ? tanhf(x) : 1.0f / (1.0f + expf(-x));
#endif
}

// Counter-based RNG for dropout masks: 2 rounds of a 64-bit mix (splitmix64 finaliser) over
// (seed, call counter, element index) -> uniform in [0,1).  Not torch's Philox stream: dropout
// parity with the CPU reference is statistical only.
__device__ __forceinline__ float rng_uniform(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

}  // namespace mfm
