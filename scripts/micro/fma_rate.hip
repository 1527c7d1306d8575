// Issue-rate microbenchmark: v_fma_f32 vs v_pk_fma_f32 vs v_fmac_f32 with a DPP operand on gfx950.
// hipcc --offload-arch=gfx950 -O3 fma_rate.hip -o fma_rate && ./fma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, float a, float b, int iters) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  f32x2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
  const f32x2 a2 = {a, a}, b2 = {b, b};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE == 0) {
        x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
        x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
      } else if (MODE == 1) {
        p0 = __builtin_elementwise_fma(p0, a2, b2); p1 = __builtin_elementwise_fma(p1, a2, b2);
        p2 = __builtin_elementwise_fma(p2, a2, b2); p3 = __builtin_elementwise_fma(p3, a2, b2);
      } else if (MODE == 3) {
        asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "+v"(x4) : "v"(x0), "v"(a));
        asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(x5) : "v"(x0), "v"(a));
        asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "+v"(x6) : "v"(x0), "v"(a));
        asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf" : "+v"(x7) : "v"(x0), "v"(a));
        asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "+v"(x1) : "v"(x0), "v"(b));
        asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(x2) : "v"(x0), "v"(b));
        asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "+v"(x3) : "v"(x0), "v"(b));
        asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf" : "+v"(p0.x) : "v"(x0), "v"(b));
      } else {
        // fmac with a quad-broadcast DPP source
        const float d0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x0), 0x00, 0xF, 0xF, false));
        const float d1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x1), 0x55, 0xF, 0xF, false));
        const float d2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x2), 0xAA, 0xF, 0xF, false));
        const float d3 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x3), 0xFF, 0xF, 0xF, false));
        x4 = fmaf(d0, a, x4); x5 = fmaf(d1, a, x5); x6 = fmaf(d2, a, x6); x7 = fmaf(d3, a, x7);
        x0 = fmaf(x4, b, x0); x1 = fmaf(x5, b, x1); x2 = fmaf(x6, b, x2); x3 = fmaf(x7, b, x3);
      }
    }
  }
  out[blockIdx.x * 1024 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
void run(const char* name, float* out) {
  const int blocks = 256 * 2, iters = 2000;
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, out, 1.0001f, 0.5f, 10);
  hipEventRecord(s);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, out, 1.0001f, 0.5f, iters);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  const double fmas = (double)blocks * 1024 * iters * 16 * 8;
  printf("%-28s %8.3f ms  %7.1f TFLOP/s  (%.2f clk per wave64 instruction-equivalent of 1 FMA/lane at 2.4 GHz, 256 CUs)\n", name, ms,
         2 * fmas / ms / 1e9, ms * 1e-3 * 2.4e9 * 256 * 4 / (fmas / 64));
}

int main() {
  float* out; hipMalloc(&out, 512 * 1024 * 4);
  run<0>("v_fma_f32", out);
  run<1>("v_pk_fma_f32", out);
  run<2>("v_mov_dpp + fma (compiler)", out);
  run<3>("v_fmac_f32_dpp (asm)", out);
  return 0;
}
