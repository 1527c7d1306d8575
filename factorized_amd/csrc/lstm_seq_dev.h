// Device-side descriptors shared by the two LSTM sequence kernel families.
#pragma once
#include "common.h"

namespace mfm {

struct SeqDev {
  float* gates; float* hs; float* cs;
  const float* w_hh; const float* w_ih; const float* b_ih; const float* b_hh;
  const float* h_init; int64_t ld_init;
  const float* dh_ext; int64_t ld_dh;
  float* d_h_init; int64_t ld_dinit;
  const float* dc_ext;
  const void* w_pack;      // bf16 path: packed weight fragments (mfm_lstm_pack_bf16), or null
  float* h_last;           // bf16 path, optional: fp32 copy of h_{T-1} [B, Hp]
  int h, Hp, hk4, is_dec, block_begin;
  int store_bf16;          // bf16 path: gates / hs / the decoders' dh_ext are __bf16 buffers (MfmSeqDesc::store_bf16)
  const float* wt_img;     // fp32 one-row BPTT, optional: this step's transposed weights in thread order (proj_role_dev.h), or null
  const float* wf_img;     // fp32 one-row forward, decoders, optional: W_ih + W_hh of this step in the forward's thread order, or null
  const float* wf1_img;    // the same for W_ih alone (the decoders' step 0), or null
};
// Transposed-weight images for the one-row BPTT kernels of the same step (lstm_seq_small.hip, small_bwd_body<.., KS = 16>):
// img4[s / 4][tid][s % 4], s = which * 4 NG + g * NG + i, holds W[g h + 16 i + (tid & 15)][2 (tid >> 4) + which] (w_ih set: W_ih + W_hh,
// the decoders' steps >= 1), zero outside the valid units.  Written by workgroups that have nothing else to do in a FORWARD
// launch of the step -- the projection role workgroups once their items are done (proj_role_dev.h), or a few blocks appended to
// the recurrence launch -- and read by the BPTT launches, which come later in the stream: nothing has to be signalled.
struct WtImgItem { const float* w_hh; const float* w_ih; float* img; int h, HKB; int fwd; };     // fwd: a forward-order image (below)
constexpr int MFM_WT_MAX = MFM_MAX_SEQ + 3;
constexpr int MFM_IMG_MAX = MFM_WT_MAX + 6;      // + the decoders' forward-order images (W_ih, W_ih + W_hh)
struct SeqLaunch {
  SeqDev d[MFM_MAX_SEQ];
  int count, T, B;
  // forward launches, optional: blocks [img_begin, img_begin + n_img_blocks) write the images of `img` and leave
  int n_img, img_begin, n_img_blocks;
  WtImgItem img[MFM_IMG_MAX];
  int bf16_dot;            // one-row forward kernels: recurrent product on bf16 dot products (MfmSeqDesc::bf16_dot on every LSTM)
};

__device__ __forceinline__ void wf_img_write(const WtImgItem* items, const int n, const int r, const int nr);
// writer r of nr (any block size); items marked `fwd` are written in the forward's register order (wf_img_write)
__device__ __forceinline__ void wt_img_write(const WtImgItem* items, const int n, const int r, const int nr) {
  const int tid = threadIdx.x, nt = blockDim.x;
#pragma unroll 1
  for (int w = 0; w < n; ++w) {
    const WtImgItem& I = items[w];
    if (I.fwd) { wf_img_write(&I, 1, r, nr); continue; }
    // (round 6: four consecutive slots of a thread side by side -- img4[(s / 4) NTH + tid][s % 4] -- so that the BPTT takes its
    // 2 NW registers with NW / 2 16-byte loads instead of 2 NW 4-byte ones; 4 NG slots per `which`: a quad never straddles it)
    const int NG = I.HKB >> 4, NTH = 8 * I.HKB, h = I.h;
    const int total = 2 * NG * NTH;
    f32x4* out = reinterpret_cast<f32x4*>(I.img);
    // 64-quad pieces (one wave, 1 KB) dealt round-robin over the WRITERS first, then over a writer's waves: the largest image
    // is 256 pieces, and dealt thread by thread they all landed on the first 16 of 128 role workgroups
    const int nwv = nt >> 6;
    for (int c = (tid >> 6) * nr + r; c * 64 < total; c += nwv * nr) {
      const int idx = c * 64 + (tid & 63);
      if (idx >= total) break;
      const int s4 = idx / NTH, t2 = idx - s4 * NTH;
      f32x4 v4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int s = 4 * s4 + e;
        const int which = s / (4 * NG), rem = s - which * 4 * NG;
        const int g = rem / NG, i = rem - g * NG;
        const int j = 16 * i + (t2 & 15), u = 2 * (t2 >> 4) + which;
        float v = 0.0f;
        if (j < h && u < h) {
          v = I.w_hh[((int64_t)g * h + j) * h + u];
          if (I.w_ih) v += I.w_ih[((int64_t)g * h + j) * h + u];
        }
        v4[e] = v;
      }
      out[idx] = v4;
    }
  }
}

// Forward-order image of a decoder's steps >= 1 weights (round 6; the launch clock, profiles/r06_launch_timeline.txt, put the
// decoders' mid-launch reload of W_ih + W_hh at 6.4 us: two strided 16-byte gathers per register quad and an add, behind step
// 0): img4[(gl * NM + m) * NTH + tid] = (W_ih + W_hh)[(2 gp + gl) h + u][16 m + 4 q .. + 3] with q = tid & 3, gp = (tid >> 2) & 1,
// u = tid >> 3, NM = HKB / 16, NTH = 8 HKB -- exactly the registers w[gl][4 m ..] of small_fwd_body<KQ, 1>, so the reload is 2 NM
// coalesced 16-byte loads per thread.  Zero outside the valid units.  Written like the transposed images (role workgroups of
// the encoder launch), read by the decoder launch behind it.  h % 4 == 0 only.  An item without w_hh: the image of W_ih alone
// (step 0: 3.0 -> 1.5 us for h = 104, the same strided gather without the add).
__device__ __forceinline__ void wf_img_write(const WtImgItem* items, const int n, const int r, const int nr) {
  const int tid = threadIdx.x, nt = blockDim.x;
#pragma unroll 1
  for (int w = 0; w < n; ++w) {
    const WtImgItem& I = items[w];
    const int NM = I.HKB >> 4, NTH = 8 * I.HKB, h = I.h;
    const int total = 2 * NM * NTH;
    f32x4* out = reinterpret_cast<f32x4*>(I.img);
    for (int idx = r * nt + tid; idx < total; idx += nr * nt) {
      const int s = idx / NTH, t2 = idx - s * NTH;
      const int gl = s / NM, m = s - gl * NM;
      const int q = t2 & 3, gp = (t2 >> 2) & 1, u = t2 >> 3;
      const int k0 = 16 * m + 4 * q;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (u < h && k0 < h) {
        const int64_t o = ((int64_t)(2 * gp + gl) * h + u) * h + k0;
        v = *reinterpret_cast<const f32x4*>(I.w_ih + o);
        if (I.w_hh) v = *reinterpret_cast<const f32x4*>(I.w_hh + o) + v;
      }
      out[idx] = v;
    }
  }
}

int seq_small_launch(SeqLaunch& L, bool bwd, hipStream_t stream);
struct LatentDev;
int seq_small_dectail_launch(SeqLaunch& L, const LatentDev& LD, const float* params, hipStream_t stream);
int seq_small_decbwd_head_launch(SeqLaunch& L, const LatentDev& LD, const float* params, float* grads, hipStream_t stream);
int seq_small_fold_launch(SeqLaunch& L, bool bwd, const LatentDev& LD, const float* params, float* grads, hipStream_t stream);
// lstm_seq_bf16.hip: bf16 MFMA operands, fp32 accumulate / cell state / saved activations
int seq_bf16_launch(SeqLaunch& L, bool bwd, hipStream_t stream);
bool bf16_seq_pays(int B);     // lstm_seq.hip: batch size from which a bf16 plan runs its recurrences on the bf16 kernels
// lstm_step.hip: step-by-step path (recurrent GEMM + cell kernel per time step) for h > MFM_SEQ_MAX_RESIDENT_H
int seq_stepwise(const MfmSeqDesc* descs, int count, int T, int B, bool bwd, hipStream_t stream);
constexpr int MFM_SEQ_MAX_RESIDENT_H = 128;

}  // namespace mfm
