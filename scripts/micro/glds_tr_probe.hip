// Probe of two gfx950 mechanisms the bf16-resident weight-gradient kernel (csrc/dw_bf16.hip) is built on:
//  (1) buffer_load_dwordx4 ... lds (LDS-DMA): lane-linear LDS destination; what an OUT-OF-RANGE lane writes (0 or nothing)
//  (2) ds_read_b64_tr_b16: which elements of a row-major [k][n] bf16 image a lane receives
// build: hipcc --offload-arch=gfx950 -O2 glds_tr_probe.hip -o glds_tr_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const unsigned short* g, int nbytes, unsigned short* out_lds, short* out_tr, int ld) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned short* l16 = (unsigned short*)lds;
  for (int i = threadIdx.x; i < 4096; i += 64) l16[i] = 0x7777;       // sentinel
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nbytes, 0x00020000);
  // lanes 0..63 request 16 bytes each at byte offset lane*16; the buffer holds only nbytes
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, threadIdx.x * 16, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 8; i += 64) out_lds[i] = l16[i];
  __syncthreads();
  // image [k][ld] bf16 at byte 2048: value = k * 256 + n
  unsigned short* img = l16 + 1024;
  for (int i = threadIdx.x; i < 16 * ld; i += 64) img[i] = (unsigned short)((i / ld) * 256 + (i % ld));
  __syncthreads();
  const int lane = threadIdx.x, bi = lane & 15, q = lane >> 4;
  // lane (bi, q): address of row (4q + bi/4), cols (bi%4)*4 .. +3  -> expect to receive column bi, rows 4q..4q+3
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + (4 * q + bi / 4) * ld + (bi % 4) * 4));
  for (int j = 0; j < 4; ++j) out_tr[lane * 4 + j] = v[j];
}
int main() {
  const int n = 64 * 8;
  std::vector<unsigned short> h(n);
  for (int i = 0; i < n; ++i) h[i] = (unsigned short)(i + 1);
  unsigned short *g, *ol; short* ot;
  hipMalloc(&g, n * 2); hipMalloc(&ol, n * 2); hipMalloc(&ot, 64 * 4 * 2);
  hipMemcpy(g, h.data(), n * 2, hipMemcpyHostToDevice);
  const int nbytes = 40 * 16;          // lanes 40..63 are out of range
  for (int ld = 16; ld <= 24; ld += 8) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 16384, 0, g, nbytes, ol, ot, ld);
    std::vector<unsigned short> o(n); std::vector<short> t(256);
    hipMemcpy(o.data(), ol, n * 2, hipMemcpyDeviceToHost); hipMemcpy(t.data(), ot, 512, hipMemcpyDeviceToHost);
    int ok_in = 1, oob_zero = 1, oob_keep = 1;
    for (int i = 0; i < 40 * 8; ++i) ok_in &= (o[i] == h[i]);
    for (int i = 40 * 8; i < n; ++i) { oob_zero &= (o[i] == 0); oob_keep &= (o[i] == 0x7777); }
    printf("ld=%d glds: in-range lane-linear %s; out-of-range lanes: %s\n", ld, ok_in ? "OK" : "MISMATCH", oob_zero ? "ZERO written" : (oob_keep ? "NOT written (sentinel kept)" : "mixed"));
    int tr_ok = 1;
    for (int lane = 0; lane < 64; ++lane)
      for (int j = 0; j < 4; ++j) { const int k = 4 * (lane >> 4) + j, nn = lane & 15; if (t[lane * 4 + j] != (short)(k * 256 + nn)) tr_ok = 0; }
    printf("ld=%d tr_b16: lane (bi,q) elem j == img[4q+j][bi]: %s\n", ld, tr_ok ? "YES" : "NO");
    if (!tr_ok) for (int lane = 0; lane < 20; ++lane) printf("  lane %d: %04x %04x %04x %04x\n", lane, (unsigned short)t[lane*4], (unsigned short)t[lane*4+1], (unsigned short)t[lane*4+2], (unsigned short)t[lane*4+3]);
  }
  return 0;
}
