// Weight images of a bf16 plan, rebuilt once per step from the fp32 master weights (round 3: one launch for all of them).
//   * the recurrences' MFMA fragment images (lstm_seq_bf16.hip, which documents the layout),
//   * the input projections' tile image + combined biases (proj_bf16.hip),
//   * the decoders' fc1 image (dec_fc1_large.hip).
// Each body maps a global thread index inside its own range to one 16-byte piece; pack_all_kernel (proj_bf16.hip) dispatches
// on the block index.  Three separate launches cost 8 + 7.5 + 5 us per step (profiles/r03_roofline_table_l_bf16.txt) for 1.7 MB.
#pragma once
#include "internal.h"

namespace mfm {

typedef __bf16 pk_bf16x8 __attribute__((ext_vector_type(8)));

// ---- recurrences
struct PackItem { const float* w_hh; const float* w_ih; pk_bf16x8* out; int h, Hp, KB, is_dec, frag_begin; };
struct PackLaunch { PackItem it[MFM_MAX_SEQ * 2]; int count; int64_t total; };

__device__ __forceinline__ void lstm_pack_body(const PackLaunch& L, int64_t gid) {
  int ii = 0;
#pragma unroll 1
  for (int i = 1; i < L.count; ++i)
    if (gid >= L.it[i].frag_begin) ii = i;
  const PackItem& it = L.it[ii];
  const int h = it.h, KB = it.KB, HKP = KB * 32;
  const int per_pack = (it.Hp >> 4) * 4 * KB * 64;
  const int npack = it.is_dec ? 4 : 2;
  int f = (int)(gid - it.frag_begin);
  if (f >= per_pack * npack) return;
  const int which = f / per_pack;
  f -= which * per_pack;
  const int lane = f & 63;
  int r = f >> 6;
  const int bi = lane & 15, q = lane >> 4;
  const bool bwd = it.is_dec ? (which >= 2) : (which == 1);
  // MODE: 0 W_hh, 1 W_ih, 2 W_ih + W_hh
  const int mode = it.is_dec ? ((which & 1) ? 1 : 2) : 0;
  // The descriptor's pointers are read ONCE, and all 16 gathered elements are requested before any is used (pad elements read
  // element 0): written as `mode == 0 ? w_hh[off] : ...` inside the element loop every element was three dependent round trips
  // (descriptor field, pointer, data -- the item index is per-thread, so the descriptor is read with vector loads) and the
  // launch that packs all weight images of a step took 8.8 us for ~1 MB.
  const float* const pa = mode == 1 ? it.w_ih : it.w_hh;
  const float* const pb = mode == 2 ? it.w_ih : pa;            // second summand (mode 2), else the same element again
  pk_bf16x8* const outp = it.out;
  int off[8];
  bool okv[8];
  if (!bwd) {
    const int kb = r % KB; r /= KB;
    const int g = r & 3, wave = r >> 2;
    const int unit = wave * 16 + bi;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kb * 32 + 8 * q + j;
      okv[j] = unit < h && k < h;
      off[j] = okv[j] ? (g * h + unit) * h + k : 0;
    }
  } else {
    const int nkb = 4 * KB;
    const int kb = r % nkb, wave = r / nkb;
    const int unit = wave * 16 + bi;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kb * 32 + 8 * q + j;
      const int g = k / HKP, up = k % HKP;
      okv[j] = unit < h && up < h;
      off[j] = okv[j] ? (g * h + up) * h + unit : 0;
    }
  }
  float xa[8], xb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { xa[j] = pa[off[j]]; xb[j] = pb[off[j]]; }
#pragma unroll
  for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(xa[j]), "+v"(xb[j]));
  pk_bf16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (__bf16)(okv[j] ? (mode == 2 ? xa[j] + xb[j] : xa[j]) : 0.0f);
  outp[(int64_t)which * per_pack + (f & ~63) + lane] = v;
}

// ---- input projections (proj_bf16.hip: tiles of [128 columns][32 k], see there)
constexpr int PJ_BN = 128, PJ_BK = 32;
constexpr int PJ_TILE = PJ_BN * PJ_BK;            // elements of a weight tile (8 KB)
struct PjPackGroup { const float* w; const float* bias; const float* bias2; int64_t ldw; int n, seg, seg_valid, k_off, k_len, kt0, nkt, tile0, nchunks, bias_off; };
struct PjPackDev { PjPackGroup g[MFM_PANEL_MAXG]; int ngroups, ntiles, nbias; __bf16* wimg; float* bimg; };

// position (in 16-byte chunks) of chunk c of tile row nn: 16 consecutive rows at one c cover the 16 slots of a 256-byte
// bank row exactly once (rows nn and nn+4 share slot group 4 (nn & 3) and are told apart by c ^ ((nn >> 2) & 3))
__host__ __device__ inline int pj_tile_slot(int nn, int c) { return nn * 4 + (c ^ ((nn >> 2) & 3)); }

__device__ __forceinline__ void proj_pack_body(const PjPackDev& L, int64_t gid) {
  const int64_t nchunk = (int64_t)L.ntiles * (PJ_TILE / 8);
  if (gid < nchunk) {
    const int t = (int)(gid / (PJ_TILE / 8));
    const int within = (int)(gid - (int64_t)t * (PJ_TILE / 8));
    const int nn = within >> 2, c = within & 3;
    int gi = 0;
#pragma unroll 1
    for (int i = 1; i < L.ngroups; ++i)
      if (t >= L.g[i].tile0) gi = i;
    const PjPackGroup& G = L.g[gi];
    // (the group's fields once, all eight elements requested together: see lstm_pack_body)
    const float* const gw = G.w;
    const int g_tile0 = G.tile0, g_nkt = G.nkt, g_kt0 = G.kt0, g_seg = G.seg, g_n = G.n, g_sv = G.seg_valid, g_koff = G.k_off, g_klen = G.k_len;
    const int64_t g_ldw = G.ldw;
    const int lt = t - g_tile0;
    const int chunk = lt / g_nkt, kt = g_kt0 + (lt - chunk * g_nkt);
    // tile row 16 fn + i of a wave's 32 rows carries the wave's column 8 (i / 4) + 4 fn + i % 4 (see the kernel's epilogue)
    const int wr = nn & 31, fn = wr >> 4, i = wr & 15;
    const int n = chunk * PJ_BN + (nn & ~31) + 8 * (i >> 2) + 4 * fn + (i & 3);
    const int sg = n / g_seg, u = n - sg * g_seg;
    const bool nok = n < g_n && u < g_sv;
    const int64_t wrow = nok ? (int64_t)(sg * g_sv + u) * g_ldw : 0;
    float x[8];
    bool okv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ko = kt * PJ_BK + c * 8 + j - g_koff;
      okv[j] = nok && ko >= 0 && ko < g_klen;
      x[j] = gw[okv[j] ? wrow + ko : 0];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(x[j]));
    pk_bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)(okv[j] ? x[j] : 0.0f);
    *reinterpret_cast<pk_bf16x8*>(L.wimg + (int64_t)t * PJ_TILE + pj_tile_slot(nn, c) * 8) = v;
    return;
  }
  const int64_t b = gid - nchunk;
  if (b >= L.nbias) return;
  int gi = 0;
#pragma unroll 1
  for (int i = 1; i < L.ngroups; ++i)
    if (b >= L.g[i].bias_off) gi = i;
  const PjPackGroup& G = L.g[gi];
  const int n = (int)(b - G.bias_off);
  const int sg = n / G.seg, u = n - sg * G.seg;
  float s = 0.0f;
  if (n < G.n && u < G.seg_valid) {
    if (G.bias) s += G.bias[sg * G.seg_valid + u];
    if (G.bias2) s += G.bias2[sg * G.seg_valid + u];
  }
  L.bimg[b] = s;
}

// ---- decoder fc1 (dec_fc1_large.hip): [NB2 * 32][FC1_LDW] bf16, rows n >= d and columns k >= h zero
constexpr int FC1_LDW = 128 + 8;      // bf16 elements per row of the W image / the H tile (pad: 16 bytes)
struct Fc1PackArgs { const float* w[3]; __bf16* out[3]; int d[3], h[3], rows[3], begin[4]; int n; };

__device__ __forceinline__ void fc1_pack_body(const Fc1PackArgs& A, int64_t gid64) {
  const int gid = (int)gid64;
  if (gid >= A.begin[A.n]) return;
  int m = 0;
#pragma unroll
  for (int i = 1; i < 3; ++i)
    if (i < A.n && gid >= A.begin[i]) m = i;
  const int idx = gid - A.begin[m];
  const int n = idx / (FC1_LDW / 8), k8 = (idx - n * (FC1_LDW / 8)) * 8;
  const float* const wm_ = m == 0 ? A.w[0] : (m == 1 ? A.w[1] : A.w[2]);
  __bf16* const om = m == 0 ? A.out[0] : (m == 1 ? A.out[1] : A.out[2]);
  const int dm = m == 0 ? A.d[0] : (m == 1 ? A.d[1] : A.d[2]), hm = m == 0 ? A.h[0] : (m == 1 ? A.h[1] : A.h[2]);
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = wm_[(n < dm && k8 + e < hm) ? (int64_t)n * hm + k8 + e : 0];
#pragma unroll
  for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(x[e]));
  pk_bf16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (__bf16)((n < dm && k8 + e < hm) ? x[e] : 0.0f);
  *reinterpret_cast<pk_bf16x8*>(om + (size_t)n * FC1_LDW + k8) = v;
}

// host side: each module fills its part, pack_all_launch (proj_bf16.hip) issues the one launch
int lstm_pack_prepare(const MfmSeqDesc* descs, int count, PackLaunch* out);                     // lstm_seq_bf16.hip
int proj_pack_prepare(const PanelLaunch& L, const ProjPlan& P, void* wimg, float* bimg, PjPackDev* out);   // proj_bf16.hip
int fc1_pack_prepare(const DecFc1LargeLaunch& L, Fc1PackArgs* out);                              // dec_fc1_large.hip
int pack_all_launch(const PackLaunch* lstm, const PjPackDev* proj, const Fc1PackArgs* fc1, hipStream_t stream);

}  // namespace mfm
