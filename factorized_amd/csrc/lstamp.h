// A clock on the LAUNCH (debug build only: make MFM_EXTRA_FLAGS=-DMFM_LAUNCH_STAMP=1, scripts/launch_timeline.sh).
//
// rocprofv3 says how long a launch took; the per-step micro-benchmarks say what a time step costs; nothing said where the
// rest of a launch goes (round 5: 78 of the 144 us the four recurrence launches of the headline step take are NOT time-loop
// work).  In a stamp build thread 0 of every workgroup of the six kernels of the B <= 32 MFM_KL_EF step stores the 100 MHz
// wall clock (s_memrealtime: one counter for the whole device, so stamps of different workgroups, XCDs and KERNELS line up on
// one axis) at a few points of its life into stamps[kernel][block][point]; scripts/launch_timeline.py runs the real step back
// to back (cold operands behind Adam included), reads the last step's stamps and prints, per launch and per workgroup role,
// entry / prologue / first and last time step / epilogue / exit, and the gaps between launches.  Points that mark "data has
// arrived" wait for the wave's outstanding loads first (wave 0 only); the cost of the instrumentation is the difference of the
// step time of the two builds, printed next to the table.  The product build compiles none of this (the macros are empty).
#pragma once
#ifndef MFM_LAUNCH_STAMP
#define MFM_LAUNCH_STAMP 0
#endif

#if MFM_LAUNCH_STAMP
#include <hip/hip_runtime.h>

namespace mfm {
constexpr int LST_KERNELS = 6;       // 0 enc fwd (foldproj)  1 dec fwd  2 fc1  3 dec BPTT  4 enc BPTT (folddw)  5 adam
constexpr int LST_BLOCKS = 512;
constexpr int LST_POINTS = 32;
namespace {
__device__ unsigned long long* g_lstamp;       // one copy per translation unit, bound by that unit's launcher
}
unsigned long long* lstamp_buffer();           // plan.hip: the one device buffer [LST_KERNELS][LST_BLOCKS][LST_POINTS]
static inline void lstamp_bind() {
  static bool done = false;
  if (done) return;
  unsigned long long* p = lstamp_buffer();
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lstamp), &p, sizeof(p));
  done = true;
}
// wave-uniform call sites only
__device__ __forceinline__ void lstamp(int kid, int point, bool wait_loads = false) {
  if (threadIdx.x < 64) {
    if (wait_loads) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && blockIdx.x < LST_BLOCKS && point < LST_POINTS)
      g_lstamp[((long)kid * LST_BLOCKS + blockIdx.x) * LST_POINTS + point] = wall_clock64();
  }
}
}  // namespace mfm
#define LSTAMP(kid, pt) ::mfm::lstamp(kid, pt)
#define LSTAMP_W(kid, pt) ::mfm::lstamp(kid, pt, true)
#define LSTAMP_BIND() ::mfm::lstamp_bind()
#else
#define LSTAMP(kid, pt) ((void)0)
#define LSTAMP_W(kid, pt) ((void)0)
#define LSTAMP_BIND() ((void)0)
#endif
