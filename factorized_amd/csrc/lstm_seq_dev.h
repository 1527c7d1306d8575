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
};
struct SeqLaunch {
  SeqDev d[MFM_MAX_SEQ];
  int count, T, B;
};

int seq_small_launch(SeqLaunch& L, bool bwd, hipStream_t stream);
struct LatentDev;
int seq_small_fold_launch(SeqLaunch& L, bool bwd, const LatentDev& LD, const float* params, float* grads, hipStream_t stream);
// lstm_seq_bf16.hip: bf16 MFMA operands, fp32 accumulate / cell state / saved activations
int seq_bf16_launch(SeqLaunch& L, bool bwd, hipStream_t stream);
bool bf16_seq_pays(int B);     // lstm_seq.hip: batch size from which a bf16 plan runs its recurrences on the bf16 kernels
// lstm_step.hip: step-by-step path (recurrent GEMM + cell kernel per time step) for h > MFM_SEQ_MAX_RESIDENT_H
int seq_stepwise(const MfmSeqDesc* descs, int count, int T, int B, bool bwd, hipStream_t stream);
constexpr int MFM_SEQ_MAX_RESIDENT_H = 128;

}  // namespace mfm
