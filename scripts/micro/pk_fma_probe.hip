// v_pk_fma_f32 with op_sel broadcast vs v_fma_f32 on gfx950: semantics (bit-exact against the scalar chain) and issue rate with
// 4 waves per SIMD, the shape of the one-row LSTM recurrences' inner product (acc[gl] += w[gl][k] * h[k], two gate rows per lane).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/pk_fma_probe.hip -o scripts/micro/pk_fma_probe && scripts/micro/pk_fma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f2 pk_fma_lo(f2 w, f2 h, f2 acc) {      // acc.{x,y} += w.{x,y} * h.x
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w), "v"(h));
  return acc;
}
__device__ __forceinline__ f2 pk_fma_hi(f2 w, f2 h, f2 acc) {      // acc.{x,y} += w.{x,y} * h.y
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(h));
  return acc;
}
constexpr int K = 32;      // k per lane
template <int MODE, bool LDS>
__global__ __launch_bounds__(1024) void probe(const float* __restrict__ w, const float* __restrict__ hin, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) float hs[K];
  const int tid = threadIdx.x;
  float w0[K], w1[K];
  f2 wp[K];
  for (int k = 0; k < K; ++k) { w0[k] = w[(tid * 2) * K + k]; w1[k] = w[(tid * 2 + 1) * K + k]; wp[k] = f2{w0[k], w1[k]}; }
  if (tid < K) hs[tid] = hin[tid];
  __syncthreads();
  float a0 = 0.f, a1 = 0.f;
  f2 acc = {0.f, 0.f}, accb = {0.f, 0.f};
  constexpr bool CH2 = MODE == 2;
  f4 hv[K / 4];
#pragma unroll
  for (int m = 0; m < K / 4; ++m) hv[m] = *reinterpret_cast<const f4*>(hs + 4 * m);
  for (int it = 0; it < iters; ++it) {
    if (LDS) {        // the recurrences' shape: the operand vector is re-read from LDS every step
#pragma unroll
      for (int m = 0; m < K / 4; ++m) hv[m] = *reinterpret_cast<const f4*>(hs + 4 * m + ((it & 1) ? 0 : 0));
      asm volatile("" ::: "memory");
    }
    if (MODE == 0) {
#pragma unroll
      for (int m = 0; m < K / 4; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) { a0 = fmaf(w0[4 * m + i], hv[m][i], a0); a1 = fmaf(w1[4 * m + i], hv[m][i], a1); }
      asm volatile("" : "+v"(a0), "+v"(a1));
    } else {
#pragma unroll
      for (int m = 0; m < K / 4; ++m) {
        const f2 lo = {hv[m][0], hv[m][1]}, hi = {hv[m][2], hv[m][3]};
        if (CH2) {      // two independent chains (not the scalar chain's summation order: timing only)
          acc = pk_fma_lo(wp[4 * m + 0], lo, acc); accb = pk_fma_hi(wp[4 * m + 1], lo, accb);
          acc = pk_fma_lo(wp[4 * m + 2], hi, acc); accb = pk_fma_hi(wp[4 * m + 3], hi, accb);
        } else {
          acc = pk_fma_lo(wp[4 * m + 0], lo, acc); acc = pk_fma_hi(wp[4 * m + 1], lo, acc);
          acc = pk_fma_lo(wp[4 * m + 2], hi, acc); acc = pk_fma_hi(wp[4 * m + 3], hi, acc);
        }
      }
      asm volatile("" : "+v"(acc), "+v"(accb));
    }
  }
  if (MODE == 0) { out[tid * 2] = a0; out[tid * 2 + 1] = a1; }
  else { out[tid * 2] = acc[0] + accb[0]; out[tid * 2 + 1] = acc[1] + accb[1]; }
}
int main() {
  const int NT = 1024, NB = 256, iters = 2000;
  std::vector<float> w(NT * 2 * K), h(K), o0(NT * 2), o1(NT * 2);
  srand(1);
  for (auto& v : w) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
  for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f);
  float *dw, *dh, *d0, *d1;
  hipMalloc(&dw, w.size() * 4); hipMalloc(&dh, h.size() * 4); hipMalloc(&d0, o0.size() * 4); hipMalloc(&d1, o1.size() * 4);
  hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dh, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float ms[2];
  for (int lds = 0; lds < 2; ++lds) {
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a);
      if (mode == 0 && lds) hipLaunchKernelGGL((probe<0, true>), dim3(NB), dim3(NT), 0, 0, dw, dh, d0, iters);
      else if (mode == 0) hipLaunchKernelGGL((probe<0, false>), dim3(NB), dim3(NT), 0, 0, dw, dh, d0, iters);
      else if (lds) hipLaunchKernelGGL((probe<1, true>), dim3(NB), dim3(NT), 0, 0, dw, dh, d1, iters);
      else hipLaunchKernelGGL((probe<1, false>), dim3(NB), dim3(NT), 0, 0, dw, dh, d1, iters);
      hipEventRecord(b); hipEventSynchronize(b);
      hipEventElapsedTime(&ms[mode], a, b);
    }
  }
  hipEventRecord(a);
  if (lds) hipLaunchKernelGGL((probe<2, true>), dim3(NB), dim3(NT), 0, 0, dw, dh, d1, iters);
  else hipLaunchKernelGGL((probe<2, false>), dim3(NB), dim3(NT), 0, 0, dw, dh, d1, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms2; hipEventElapsedTime(&ms2, a, b);
  printf("   two independent v_pk_fma_f32 chains: %.1f clk\n", ms2 * 1e-3 * 2.4e9 / iters);
  printf("operand vector %s: v_fma_f32 %.1f clk, v_pk_fma_f32 %.1f clk per iteration (2 K = %d FMA per lane, 16 waves per CU, 2.4 GHz assumed)\n",
         lds ? "re-read from LDS every iteration (8 ds_read_b128)" : "in registers", ms[0] * 1e-3 * 2.4e9 / iters, ms[1] * 1e-3 * 2.4e9 / iters, 2 * K);
  }
  hipMemcpy(o0.data(), d0, o0.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(o1.data(), d1, o1.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (size_t i = 0; i < o0.size(); ++i) bad += (o0[i] != o1[i]);
  const double fl = 2.0 * 2 * K * iters;      // per lane
  printf("v_fma_f32   : %.3f ms  %.1f clk per (2 K = %d FMA) iteration and wave-quad at 2.4 GHz\n", ms[0], ms[0] * 1e-3 * 2.4e9 / iters, 2 * K);
  printf("v_pk_fma_f32: %.3f ms  %.1f clk per iteration; ratio %.2f; %d of %zu results differ (bit-exact expected)\n", ms[1], ms[1] * 1e-3 * 2.4e9 / iters,
         ms[0] / ms[1], bad, o0.size());
  (void)fl;
  return bad != 0;
}
