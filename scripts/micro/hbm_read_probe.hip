// Achievable HBM READ rate on one MI355X, by access form (round 5).  Why: every kernel of the large-batch step reads at
// 1.7 - 2.5 TB/s (rocprofv3 FETCH_SIZE / duration) whatever its structure; is that this code or the part?
//   form 0: global_load_dwordx4 into registers, U loads in flight per thread, W workgroups per CU
//   form 1: global_load_lds_dwordx4 (LDS-DMA), one 512-thread workgroup per CU, S stages of C KB in flight (dw_stream's form)
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/hbm_read_probe.hip -o /tmp/hbm_read_probe ; run: /tmp/hbm_read_probe [MB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

template <int U>
__global__ __launch_bounds__(256) void read_regs(const f32x4* __restrict__ p, size_t n16, float* out) {
  // workgroup b streams a contiguous range; thread t reads piece (i * 256 + t) of it
  const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
  const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n16 ? b0 + per : n16;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  size_t i = b0 + threadIdx.x;
  for (; i + (size_t)(U - 1) * 256 < b1; i += (size_t)U * 256) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + i + (size_t)u * 256);
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  for (; i < b1; i += 256) acc += p[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) out[0] = acc[0];
}

// LDS-DMA: every thread issues NI instructions per stage (1 KB per wave and instruction), S stages in flight
template <int NI>
__global__ __launch_bounds__(512) void read_dma(const unsigned char* __restrict__ p, size_t bytes, int S, float* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t stage_bytes = (size_t)NI * 512 * 16;
  const size_t per = (bytes / gridDim.x) / stage_bytes * stage_bytes;
  const unsigned char* base = p + (size_t)blockIdx.x * per;
  const int n_chunks = (int)(per / stage_bytes);
  auto issue = [&](int chunk, int stage) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const unsigned char* g = base + (size_t)chunk * stage_bytes + ((size_t)i * 512 + tid) * 16;
      const unsigned ldsaddr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void*)(dsm + stage * stage_bytes + i * 512 * 16 + wave * 64 * 16));
      unsigned m0_saved;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(m0_saved) : "v"(g), "s"(ldsaddr) : "memory");
    }
  };
  for (int k = 0; k < S - 1; ++k) issue(k < n_chunks ? k : 0, k);
  int stage = 0, nxt = S - 1;
  float acc = 0.f;
  for (int c = 0; c < n_chunks; ++c) {
    // (the probe waits for everything but the youngest stage group: S <= 3 exact, deeper S slightly pessimistic)
    if (S == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (NI == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (NI == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    issue(c + S - 1 < n_chunks ? c + S - 1 : 0, nxt);
    acc += *reinterpret_cast<const float*>(dsm + stage * stage_bytes + tid * 16);
    nxt = stage;
    stage = stage + 1 == S ? 0 : stage + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 1.2345f) out[0] = acc;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; hipEventElapsedTime(&ms, a, b); return ms; }

int main(int argc, char** argv) {
  const size_t mb = argc > 1 ? atol(argv[1]) : 1024;
  const size_t bytes = mb << 20;
  unsigned char* buf; float* out;
  hipMalloc(&buf, bytes); hipMalloc(&out, 16);
  hipMemset(buf, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int cus = 256;
  auto report = [&](const char* name, float ms, int reps) {
    printf("%-44s %8.1f us  %7.0f GB/s\n", name, 1e3 * ms / reps, bytes / 1e9 / (ms / reps / 1e3));
  };
  const int reps = 5;
  for (int w : {1, 2, 4, 8, 16, 32}) {
    char nm[96];
#define RUN_REGS(U) { read_regs<U><<<cus * w, 256>>>((const f32x4*)buf, bytes / 16, out); hipDeviceSynchronize(); hipEventRecord(e0); \
      for (int r = 0; r < reps; ++r) read_regs<U><<<cus * w, 256>>>((const f32x4*)buf, bytes / 16, out); hipEventRecord(e1); hipEventSynchronize(e1); \
      snprintf(nm, sizeof nm, "regs: %2d WG/CU x 256 thr, %d loads in flight", w, U); report(nm, time_ms(e0, e1), reps); }
    RUN_REGS(4) RUN_REGS(8)
  }
  hipFuncSetAttribute((const void*)read_dma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)read_dma<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)read_dma<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int S : {2, 3}) {
    char nm[96];
#define RUN_DMA(NI, G) { const size_t sm = (size_t)S * NI * 512 * 16; if (sm <= 160 * 1024 / (G > cus ? 2 : 1)) { \
      read_dma<NI><<<G, 512, sm>>>(buf, bytes, S, out); hipDeviceSynchronize(); hipEventRecord(e0); \
      for (int r = 0; r < reps; ++r) read_dma<NI><<<G, 512, sm>>>(buf, bytes, S, out); hipEventRecord(e1); hipEventSynchronize(e1); \
      snprintf(nm, sizeof nm, "lds-dma: %d WG, %d stages x %d KB", G, S, NI * 8); report(nm, time_ms(e0, e1), reps); } }
    RUN_DMA(2, cus) RUN_DMA(4, cus) RUN_DMA(6, cus) RUN_DMA(2, 2 * cus) RUN_DMA(4, 2 * cus)
  }
  return 0;
}
