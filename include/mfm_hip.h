/*
 * mfm_hip.h -- C ABI of libmfm_hip.so: the MI355X (gfx950 / CDNA4) implementation of the
 * MFM training hot path of pliang279/factorized.
 *
 * The reference has no FFI / plugin boundary: it is pure Python and all arithmetic lives in
 * PyTorch ops (nn.LSTMCell, nn.Linear, autograd, optim.Adam).  The drop-in boundary is
 * therefore the Python class surface (factorized_amd/mfm_model.py mirrors mfm_model.py); THIS
 * header is the native boundary underneath it -- what a maintainer of the reference would bind
 * with ctypes.  Each entry point names the reference code it replaces
 * (file:line into the reference repo).
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer unless it says host.
 *   - caller owns every buffer (PyTorch caching allocator); nothing here allocates device memory
 *     that outlives a call.  Workspace sizes come from mfm_plan_workspace_bytes().
 *   - all work is enqueued on the hipStream_t passed as `stream` (void*, 0 = default stream) and
 *     the call returns immediately; no hidden synchronisation (the reference only syncs at
 *     `.item()`, mfm_mosi.py:442).
 *   - return 0 on success, negative MFM_ERR_* otherwise; mfm_last_error() gives the text.
 *     No C++ exception crosses this boundary.
 *   - fp32 everywhere; LSTM gate order is torch's i,f,g,o; weight layouts are torch's
 *     (weight_ih [4h,d], weight_hh [4h,h], Linear.weight [out,in]), row-major.
 *   - "padded hidden" Hp = round_up(h,16).  Sequence state buffers use the padded layouts
 *       gates [T,B,4,Hp]   hs [T,B,Hp]   cs [T,B,Hp]
 *     pad units hold exact zeros for h/c (0.5/0/0.5 gates); pad rows do not exist (B is exact).
 */
#ifndef MFM_HIP_H_
#define MFM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFM_OK 0
#define MFM_ERR_ARG (-1)
#define MFM_ERR_HIP (-2)
#define MFM_ERR_UNSUPPORTED (-3)

#define MFM_ABI_VERSION 4

int mfm_abi_version(void);
const char* mfm_last_error(void);
/* number of CUs of the current device (for grid heuristics / reporting). */
int mfm_device_cus(void);

/* ------------------------------------------------------------------------------------------
 * Grouped fp32 GEMM on the f32 MFMA (v_mfma_f32_16x16x4_f32), exact fp32 FMA chains.
 *   for z in [0,batch):  C[z][m][n] (+)= alpha * sum_k A(z,m,k) * B(z,k,n) + bias[z][n] + bias2[z][n]
 * A(z,m,k) = a[z*a_sz + m*a_sm + k*a_sk], B(z,k,n) = b[z*b_sz + k*b_sk + n*b_sn],
 * C element at c[z*c_sz + m*ldc + n].  Columns n in [n_valid, n) are produced as exact zeros
 * (pad units).  accumulate!=0 -> atomic += into c (and c2 if non-null), needed for split_k>1.
 * Replaces every nn.Linear / LSTMCell addmm and its AddmmBackward on the path:
 * mfm_model.py:56,83,85 (x W_ih^T hoisted over all T), :61,90 (fc1), autograd of the same.
 */
typedef struct MfmGemmDesc {
  const float* a; const float* b; float* c; float* c2;
  const float* bias; const float* bias2;
  int64_t a_sz, a_sm, a_sk;
  int64_t b_sz, b_sk, b_sn;
  int64_t c_sz, ldc, bias_sz;
  int32_t m, n, k, n_valid, batch, split_k, accumulate;
  float alpha;
  /* bf16-RESIDENT operands (ABI 2; mfm_gemm_grouped_bf16 only, 0 = the fp32 buffers described above).  A bf16 plan keeps
   * its saved activations in HBM as bf16 (plan.hip): a_bf16 -> `a` points to __bf16 elements (strides count bf16
   * elements; the unit-stride axis must start on 16-byte boundaries), loaded straight into the LDS image without a
   * rounding pass; c_bf16 -> `c` (and c2) receive bf16 (nearest even) instead of fp32 -- plain, non-accumulating
   * products only. */
  int32_t a_bf16, c_bf16;
  int32_t reserved_[2];   /* MUST be zero (memset the struct first): used inside the library; the entry points reject anything else */
} MfmGemmDesc;

int mfm_gemm_grouped_f32(const MfmGemmDesc* descs /*host*/, int count, void* stream);
/* The same products with bf16 MFMA operands (v_mfma_f32_16x16x32_bf16) and fp32 accumulation: a and b are still
 * fp32 buffers, rounded to bf16 (nearest even) on the way from the global tile to LDS; c, bias, alpha, the
 * epilogue and the atomics are fp32 exactly as above.  "bf16 compute, fp32 master weights" of BASELINE.json
 * configs 2-4.  A group holding an operand that is not unit-stride along m/n or k runs on the fp32 kernel. */
int mfm_gemm_grouped_bf16(const MfmGemmDesc* descs /*host*/, int count, void* stream);

/* ------------------------------------------------------------------------------------------
 * Whole-sequence LSTM recurrence (one persistent workgroup per 16 batch rows per LSTM; weights
 * stay in VGPRs for all T steps, h is exchanged through LDS, c never leaves registers).
 * Up to MFM_MAX_SEQ independent LSTMs run in ONE launch.
 *
 * Encoder form (is_dec=0), replaces encoderLSTM.forward's time loop mfm_model.py:47-58:
 *   in : gates = x_t W_ih^T + b_ih + b_hh for all t (from mfm_gemm_grouped_f32), w_hh
 *   out: gates <- activated (i,f,g,o), hs, cs.
 * Decoder form (is_dec=1), replaces decoderLSTM.forward's loop mfm_model.py:72-88:
 *   step 0 consumes h_init [B, ld_init] through w_ih from zero state; steps >=1 feed the
 *   hidden state back as the input, i.e. gates = h (W_ih+W_hh)^T + b_ih + b_hh.
 */
#define MFM_MAX_SEQ 6
typedef struct MfmSeqDesc {
  float* gates; float* hs; float* cs;
  const float* w_hh; const float* w_ih; const float* b_ih; const float* b_hh;
  const float* h_init; int64_t ld_init;
  /* backward only */
  const float* dh_ext;   /* dec: [T,B,Hp] grad wrt every h_t; enc: [B, ld_dh] grad wrt h_{T-1} */
  int64_t ld_dh;
  float* d_h_init;       /* dec: [B, ld_dinit] out (grad wrt h_init); enc: unused */
  int64_t ld_dinit;
  int32_t h, is_dec;
  const float* dc_ext;   /* optional [T,B,Hp]: external grad wrt every CELL state c_t (the MFN encoder
                            reads c_t, mfm_model.py:171-173); NULL for the plain encoders/decoders */
  void* w_pack;          /* bf16 entry points only, optional: mfm_lstm_pack_bytes(h, is_dec) bytes filled by
                            mfm_lstm_pack_bf16 from the CURRENT weights (bf16 MFMA fragments in lane order); NULL =
                            the recurrence kernels gather their fragments from the fp32 matrices themselves (slow) */
  /* ABI 2, bf16 entry points only: bf16-RESIDENT saved activations.  store_bf16 != 0 -> `gates` (x-projection in,
     activated gates / dA out), `hs` and the decoders' per-step `dh_ext` [T,B,Hp] are __bf16 buffers of the same shapes
     (half the bytes of every time step; the cell state `cs`, dc_ext, the encoders' dh_ext [B, ld_dh], h_init and
     d_h_init stay fp32).  The recurrent product rounds h_{t-1} / dA_t to bf16 anyway, so what is stored is exactly
     what the MFMA consumed; the x-projection and the saved gate activations are rounded once more (nearest even). */
  int32_t store_bf16;
  int32_t bf16_dot;      /* ABI 3, fp32 entry points, forward, small batches (one-row workgroups): the recurrent product runs on
                            bf16 dot products (v_dot2c_f32_bf16; W and h_{t-1} rounded to bf16, fp32 accumulation) -- the small-batch
                            counterpart of the bf16 entry points.  Honoured when every LSTM of the call sets it and the launcher
                            has the variant (the decoders of the canonical sizes); otherwise the fp32 product runs. */
  float* h_last;         /* optional [B, Hp] fp32: forward also writes h_{T-1} here (the latent stack's input stays fp32) */
} MfmSeqDesc;

int mfm_lstm_seq_fwd(const MfmSeqDesc* descs /*host*/, int count, int T, int B, void* stream);

/* bf16 variants: W (rounded once per launch; the decoder's W_ih + W_hh summed in fp32 first), h_{t-1} and dA_t
 * enter v_mfma_f32_16x16x32_bf16 as bf16, accumulation / gate math / cell state / every saved tensor stay fp32.
 * One kernel family for all batch sizes (csrc/lstm_seq_bf16.hip); h > 128 falls back to the fp32 step-by-step path. */
int64_t mfm_lstm_pack_bytes(int32_t h, int32_t is_dec);
int mfm_lstm_pack_bf16(const MfmSeqDesc* descs /*host*/, int count, void* stream);
int mfm_lstm_seq_fwd_bf16(const MfmSeqDesc* descs /*host*/, int count, int T, int B, void* stream);
int mfm_lstm_seq_bwd_bf16(const MfmSeqDesc* descs /*host*/, int count, int T, int B, void* stream);

/* BPTT over the saved gates/cs (autograd of the loops above).  On return `gates` holds the
 * pre-activation gate gradients dA[T,B,4,Hp] (in place); weight/bias gradients are then plain
 * GEMMs over dA (see plan.hip).  For decoders d_h_init receives dA_0 W_ih. */
int mfm_lstm_seq_bwd(const MfmSeqDesc* descs /*host*/, int count, int T, int B, void* stream);

/* ------------------------------------------------------------------------------------------
 * bf16-RESIDENT plans: all weight gradients of ONE LSTM from bf16 buffers in one pass over the gate gradients
 * (csrc/dw_bf16.hip; what autograd accumulates into weight_ih / weight_hh / bias_ih / bias_hh of an nn.LSTMCell unrolled
 * over T, reference mfm_model.py:56,83,85):
 *   dw_ih [4h, dx] += sum_r dA[r]^T xb[r, :dx]      dw_hh [4h, h] (and dw_hh2) += sum_{r >= shift} dA[r]^T hs[r - shift]
 *   db_ih, db_hh [4h] += sum_r dA[r]
 * dA [rows, 4, Hp] bf16 (pad units zero), xb [rows, ldx] bf16 with ldx >= round_up(dx, 16) (NULL: no input product --
 * the decoders, whose step input is h_{t-1}: pass dw_hh2 = dW_ih), hs [rows, Hp] bf16, shift = B (one time step).
 * Test / tuning entry point; the fused plan builds the same launch for all LSTMs and the decoders' fc1 at once. */
int mfm_dw_bf16_lstm(const void* dA, int32_t rows, int32_t h, const void* xb, int32_t ldx, int32_t dx, const void* hs,
                     int32_t shift, float* dw_ih, float* dw_hh, float* dw_hh2, float* db_ih, float* db_hh, void* stream);
/* The same one pass over fp32 buffers (fp32 plans at large T*B): dA [rows, 4, Hp], hs [rows, Hp] fp32; x is the batch itself,
 * [rows, ldx] fp32, of which columns [xcol0, xcol0 + dx) are this LSTM's input (any dword-aligned column range). */
int mfm_dw_f32_lstm(const float* dA, int32_t rows, int32_t h, const float* x, int32_t ldx, int32_t xcol0, int32_t dx,
                    const float* hs, int32_t shift, float* dw_ih, float* dw_hh, float* dw_hh2, float* db_ih, float* db_hh,
                    void* stream);

/* ------------------------------------------------------------------------------------------
 * Reconstruction loss + gradient, replaces nn.MSELoss over x_hat vs the input slices and its
 * backward (mfm_mosi.py:412,437):  loss_slot += sum((xhat-x)^2) * inv_count (un-weighted mean),
 * dxhat = grad_scale * (xhat - x).  x is a column slice of the [T,B,D] batch (row stride ldx).
 */
int mfm_mse_fwd_bwd(const float* xhat, const float* x, int64_t ldx, int64_t rows, int32_t d,
                    float inv_count, float grad_scale, float* dxhat, float* loss_slot, void* stream);

/* ------------------------------------------------------------------------------------------
 * Memory Fusion Network: the memory recurrence of MFN.forward (reference mfm_model.py:177-181),
 *   gamma_n = sigmoid(gamma_n_fc2(drop(relu(gamma_n_fc1([attended_t, mem])))))  n = 1,2
 *   mem     = gamma1 * mem + gamma2 * cHat_t
 * for all T steps in one launch (one workgroup per batch row).  gamma_n_fc1 is split by columns:
 * a1/a2 hold its attended part INCLUDING the bias for all t (a grouped GEMM over [T*B, .]), w1m/w2m
 * are its memory columns [H_n, M] (row-major, contiguous).  Forward overwrites a1/a2 with the
 * post-relu/dropout activations and saves gamma_n and mem_t; backward turns gam1/gam2 into the
 * pre-sigmoid gradients dz_n in place and writes du_n (gradient wrt a_n's pre-activation, which is also
 * the gradient wrt the precomputed attended part) and dchat.  Weight gradients are outer-product sums
 * of these tensors (grouped GEMMs, factorized_amd/mfm_model.py).
 * Returns MFM_ERR_UNSUPPORTED when M / H_n do not fit the register-resident layout. */
typedef struct MfmMemDesc {
  float* a1; float* a2;                 /* [T,B,H1], [T,B,H2] */
  const float* chat;                    /* [T,B,M] */
  const float* w1m; const float* w2m;   /* [H1,M], [H2,M] */
  const float* w1b; const float* b1b;   /* gamma1_fc2: [M,H1], [M] */
  const float* w2b; const float* b2b;   /* gamma2_fc2: [M,H2], [M] */
  float* gam1; float* gam2; float* mems;   /* [T,B,M] each */
  float* mem_out;                       /* [B,M] last memory (forward) */
  const float* dmem_out;                /* [B,M] dL/d mem_T (backward) */
  float* du1; float* du2; float* dchat; /* backward outputs [T,B,H1], [T,B,H2], [T,B,M] */
  int32_t T, B, M, H1, H2, train;
  float p1, p2;                         /* dropout probabilities of gamma1/gamma2 */
  uint64_t seed;
  const uint64_t* seed_dev;             /* optional device word added to `seed` when the kernel runs: lets a captured
                                           hipGraph draw new masks on every replay (the caller advances it in-graph) */
  int64_t ld_wm;                        /* row stride of w1m / w2m (0 = M): lets them be the memory COLUMNS of the
                                           reference's gamma_n_fc1.weight [H_n, att+M] in place, without a copy */
  int32_t dchat_pre_tanh;               /* backward: write dchat * (1 - chat^2), the gradient wrt the PRE-activation of
                                           cHat = tanh(.) (chat then holds the tanh output), instead of dL/dchat */
  int32_t reserved_;
} MfmMemDesc;

int mfm_mfn_mem_fwd(const MfmMemDesc* desc /*host*/, void* stream);
int mfm_mfn_mem_bwd(const MfmMemDesc* desc /*host*/, void* stream);

/* ------------------------------------------------------------------------------------------
 * MMD regulariser of the non-KL MFM, replaces loss_MMD / compute_kernel (reference mfm_model.py:14-34):
 *   *loss += mean K(g,g) + mean K(z,z) - 2 mean K(g,z),  K(a,b) = exp(-mean((a-b)^2)/dim)
 * z, gauss [B, dim] contiguous (the reference draws gauss ~ N(0,1) on the host; the caller supplies it);
 * dz (optional) receives d mmd / d z.  dim <= 256. */
int mfm_mmd_fwd_bwd(const float* z, const float* gauss, int32_t B, int32_t dim, float* loss /*accumulated*/,
                    float* dz /*[B,dim] or NULL*/, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused Adam on one flat parameter buffer (torch.optim.Adam defaults semantics,
 * mfm_mosi.py:403,441): m,v,p updated in place; g is multiplied by grad_scale first (DP
 * averaging).  `step` is the 1-based step count used for bias correction.
 */
int mfm_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, int32_t step, float lr,
                  float beta1, float beta2, float eps, float grad_scale, void* stream);

/* The same update restricted to up to MFM_ADAM_MAX_SPANS disjoint element ranges of the flat buffer, each with its
 * own 1-based step count; elements outside every span keep p, m and v untouched.  This is torch.optim.Adam's
 * treatment of parameters whose .grad is None (skipped; per-parameter step counters), which staged training
 * relies on: train_beta_vae (reference mfm_mosi.py:278-281, 346-358) trains gen+reg first -- the classifier gets no
 * gradient -- then disc+reg -- the decoders and the modality z->f MLPs get none.  begin/end are multiples of 4. */
/* Guarded forms (ABI 3): `guard` is an optional device pointer to ONE float -- by convention a spare element of the gradient
 * buffer itself, so that it travels through the data-parallel all-reduce with the gradients.  The kernels read it first and
 * leave p, m and v untouched unless it is exactly 0.0f.  The fused plan stores a NaN there when an in-launch hand-over of the
 * step gave up waiting (plan option "grad_guard_offset", below): a step whose gradients cannot be trusted does not reach the
 * parameters.  guard == NULL: the unguarded update above. */
int mfm_adam_flat_guarded(float* p, const float* g, float* m, float* v, int64_t n, int32_t step, float lr,
                          float beta1, float beta2, float eps, float grad_scale, const float* guard, void* stream);

/* Capturable form (ABI 3): the 0-based count of updates applied so far and the learning rate live in device memory
 * (`step_dev`, `lr_dev`); the bias corrections are formed on the device and a one-thread launch behind the update advances the
 * counter (not when the guard skipped the step).  This is what a hipGraph of a whole training step replays: kernel arguments
 * are frozen at capture, device words are not (torch.optim.Adam(capturable=True) semantics).  n: a multiple of 4. */
int mfm_adam_flat_dev(float* p, const float* g, float* m, float* v, int64_t n, int32_t* step_dev, const float* lr_dev,
                      float beta1, float beta2, float eps, float grad_scale, const float* guard, void* stream);

#define MFM_ADAM_MAX_SPANS 8
typedef struct MfmAdamSpan { int64_t begin, end; int32_t step; int32_t reserved; } MfmAdamSpan;
int mfm_adam_flat_spans(float* p, const float* g, float* m, float* v, const MfmAdamSpan* spans /*host*/, int32_t nspans,
                        float lr, float beta1, float beta2, float eps, float grad_scale, void* stream);
int mfm_adam_flat_spans_guarded(float* p, const float* g, float* m, float* v, const MfmAdamSpan* spans /*host*/, int32_t nspans,
                                float lr, float beta1, float beta2, float eps, float grad_scale, const float* guard, void* stream);

/* ------------------------------------------------------------------------------------------
 * Gradient all-reduce of the data-parallel step (the reference has no multi-GPU
 * path): in-place fp32 sum of one flat buffer over all ranks of one node, ONE kernel launch on the
 * caller's stream, no host synchronisation.  Ranks are one process per GPU; every rank owns an uncached
 * staging block that its peers map through HIP IPC and write over xGMI (two-shot: scatter-reduce + gather,
 * csrc/p2p.hip).  Every slice is summed once, by its owner, in rank order: all ranks receive bit-identical
 * results.  Set-up: create on every rank -> export 64-byte handle -> exchange the handles out of band
 * (factorized_amd/comm.py uses torch.distributed.all_gather_object) -> connect -> barrier.
 * Waits inside the kernel are bounded (MFM_P2P_TIMEOUT_MS, default 10000): a missing peer raises the
 * flag read by mfm_p2p_status instead of hanging the device. */
int mfm_p2p_create(int32_t nranks /*<=8*/, int32_t rank, int64_t max_elems, void** handle);
int mfm_p2p_handle_bytes(void);
int mfm_p2p_export(void* handle, void* out /*mfm_p2p_handle_bytes()*/);
int mfm_p2p_connect(void* handle, const void* all_handles /*nranks x mfm_p2p_handle_bytes(), rank order*/);
/* ranks that live in ONE process (one host thread / stream per GPU with peer access enabled, or several
 * streams of one GPU in tests) skip IPC: hand every rank the others' local base pointers */
void* mfm_p2p_local_base(void* handle);
int mfm_p2p_connect_bases(void* handle, const void* const* bases /*nranks, rank order; own entry ignored*/);
int mfm_p2p_allreduce(void* handle, float* buf /*16-byte aligned*/, int64_t n /*<= max_elems*/, void* stream);
/* the same exchange with the flat Adam update (mfm_adam_flat semantics) applied by the rank-local copy of the
 * kernel as each reduced slice arrives: all-reduce + optimizer in one launch.  `grads` still receives the sum. */
int mfm_p2p_allreduce_adam(void* handle, float* grads, float* p, float* m, float* v, int64_t n, int32_t step,
                           float lr, float beta1, float beta2, float eps, float grad_scale, void* stream);
/* ABI 3: the same with a guard word (mfm_adam_flat_guarded): element `guard_index` of `grads` (-1: none).  A rank whose guard
 * is raised says so in its first-push flags; every rank then completes the exchange (the sum carries the NaN too) but NO rank
 * touches p, m, v -- the replicas skip the same steps and stay bit-identical. */
int mfm_p2p_allreduce_adam_guarded(void* handle, float* grads, float* p, float* m, float* v, int64_t n, int32_t step,
                                   float lr, float beta1, float beta2, float eps, float grad_scale, int64_t guard_index,
                                   void* stream);
int mfm_p2p_status(void* handle, int32_t* timed_out /*1 if any wait gave up since create (synchronises)*/);
/* diagnosis of a sub-par scaling run: ticks (100 MHz wall clock) workgroup 0 of THIS rank spent spinning in flag round 1 (the
 * peers' first pushes) and round 2 (the owners' results), summed over the exchanges since create / the last reset, and the
 * number of exchanges: out[0], out[1], out[2] (synchronises; bench.py --gpus N prints them per rank) */
int mfm_p2p_wait_stats(void* handle, int64_t* out /*[3]*/, int32_t reset);
void mfm_p2p_destroy(void* handle);

/* ------------------------------------------------------------------------------------------
 * The fused MFM_KL_EF training / inference plan (mfm_model.py:557-660 + mfm_mosi.py:424-442).
 * One call enqueues the whole step: input projections -> 4 encoder recurrences -> latent
 * heads/MLPs/classifier + KLD + L1|CE -> 3 decoder recurrences -> fc1 + MSE -> full backward
 * -> (optional) Adam.  Parameters live in ONE flat fp32 buffer; `param_offsets` gives the
 * element offset of each of the MFM_KLEF_NPARAM tensors in reference state_dict order
 * (encoder_l.lstm.weight_ih ... fy_to_y_fc2.bias; 78 tensors).
 */
#define MFM_KLEF_NPARAM 78
#define MFM_LOSS_SLOTS 8  /* [0]=disc [1]=mse_l [2]=mse_a [3]=mse_v [4]=kld [5]=total loss */

typedef struct MfmPlanConfig {
  int32_t d_l, d_a, d_v;
  int32_t zl, za, zv, zy;
  int32_t fl, fa, fv, fy;
  int32_t output_dim;
  int32_t loss_kind;      /* 0 = L1 (mfm_mosi.py:411,438), 1 = cross-entropy (mfm_you.py:451,484) */
  int32_t T, B;
  float lda_xl, lda_xa, lda_xv, lda_reg;   /* mfm_mosi.py:433,437 */
  float drop_zy, drop_zl, drop_za, drop_zv, drop_y;
  float reg_scale;        /* extra factor on the KLD term (DP: world size) */
  int32_t precision;      /* 0 = fp32 everywhere (BASELINE config 1; 1e-4 parity with the reference);
                             1 = bf16 MFMA operands in every GEMM and LSTM recurrence, fp32 accumulation, master
                                 weights, Adam moments, cell state, saved activations, latent stack and losses
                                 (BASELINE configs 2-4; gate: matched loss curve) */
  /* ---- model variant (the classes train_mfm picks by config['type'], reference mfm_mosi.py:398-401) */
  int32_t variant;        /* 0 = MFM_KL_EF (early-fusion LSTM for z_y; mfm_model.py:557-660)
                             1 = MFM_KL    (Memory Fusion Network for z_y + KLD;  mfm_model.py:662-764, 93-199)
                             2 = MFM       (MFN for z_y, no logvar heads, MMD regulariser; mfm_model.py:469-555, 14-34) */
  int32_t hl, ha, hv;     /* variants 1, 2: MFN LSTMCell sizes (config["h_dims"]) */
  int32_t mem_dim;        /* memsize; windowsize is 2 (cStar = [c_{t-1}, c_t]) */
  int32_t nn1, nn2, g1, g2;                       /* hidden widths of att1 / att2 / gamma1 / gamma2 (*Config["shapes"]) */
  float drop_nn1, drop_nn2, drop_g1, drop_g2;     /* their dropouts (*Config["drop"]) */
} MfmPlanConfig;

typedef struct MfmPlan MfmPlan;

/* number of parameter tensors of a variant in reference state_dict order: 78 / 104 / 90 */
int mfm_plan_num_params(int32_t variant);
/* host-side: builds the op tables; no device allocation.  param_offsets has mfm_plan_num_params(cfg->variant)
 * entries: the element offset of every tensor of the reference model's state_dict(), in its order, inside the
 * flat buffer (the MFN's out_fc1 / out_fc2 exist in the state_dict but are unused by forward, as in the reference). */
int mfm_plan_create(const MfmPlanConfig* cfg, const int64_t* param_offsets /*[mfm_plan_num_params(variant)]*/,
                    int64_t n_params_total, MfmPlan** out);
/* variant 2 only: the N(0,1) sample loss_MMD draws for every forward (reference mfm_model.py:26: torch.randn on the
 * host).  [B, zl+za+zv+zy] row-major device buffer read by the next forward / step calls; the caller refreshes it
 * (or keeps it fixed for parity runs). */
int mfm_plan_set_gauss(MfmPlan* plan, const float* gauss);
void mfm_plan_destroy(MfmPlan* plan);
int64_t mfm_plan_workspace_bytes(const MfmPlan* plan);
/* byte offset inside the workspace of 64 uint64 shader-clock stamps the latent kernels write when
 * the environment variable MFM_LATENT_DBG is set (kernel tuning aid, scripts/latent_phases.py). */
int64_t mfm_plan_debug_offset(const MfmPlan* plan);

/* zero the workspace and write its constant regions (a ones vector used for bias-gradient
 * column sums).  Call once after allocating `workspace` (and again if it is re-allocated). */
int mfm_plan_init_workspace(MfmPlan* plan, void* workspace, void* stream);

/* Forward only (evaluate/predict, mfm_mosi.py:445-465; also the first half of a step).
 * x [T,B,D] time-major contiguous, y [B] (L1, output_dim 1) / [B,output_dim] (L1) / int64 [B] (CE).
 * train!=0 applies dropout with the counter-based generator keyed by (seed, call counter).
 * Outputs (any may be NULL): xhat_l/a/v [T,B,d_*], y_hat [B,output_dim], losses[MFM_LOSS_SLOTS]. */
int mfm_plan_forward(MfmPlan* plan, const float* params, const float* x, const void* y, int train,
                     uint64_t seed, void* workspace, float* xhat_l, float* xhat_a, float* xhat_v,
                     float* y_hat, float* losses, void* stream);

/* Backward of the last mfm_plan_forward on the same workspace: fills grads (same flat layout,
 * zeroed first).  stage: 0 joint loss (mfm_mosi.py:439), 1 gen+reg, 2 disc+reg (train_beta_vae,
 * mfm_mosi.py:278-281). */
int mfm_plan_backward(MfmPlan* plan, const float* params, const float* x, const void* y, int stage,
                      void* workspace, float* grads, void* stream);

/* Backward of the last mfm_plan_forward for ARBITRARY upstream gradients (the autograd path of the
 * nn.Module mirror: the caller's own loss.backward(), mfm_mosi.py:440): d_xhat_* [T,B,d_*],
 * d_yhat [B,output_dim], d_reg = device scalar dLoss/dKLD. */
int mfm_plan_backward_ext(MfmPlan* plan, const float* params, const float* x, const float* d_xhat_l,
                          const float* d_xhat_a, const float* d_xhat_v, const float* d_yhat,
                          const float* d_reg, void* workspace, float* grads, void* stream);

/* ---- the module path's LAZY losses (round 5).  The reference's loop (mfm_mosi.py:427-441) calls model.forward(batch_X), then
 * builds  loss = L1(y_hat, y) + sum_m lda_m MSE(x_hat_m, x_m) + lda_mmd * reg  from the outputs and calls loss.backward().
 * The plan's forward already holds the three MSE terms, the regulariser and lda_m-scaled d x_hat (decoder fc1 epilogue): when
 * the loop's loss is exactly such a weighted sum (factorized_amd/lazy.py recognises it), no x_hat tensor, no torch loss kernel
 * and no autograd graph is needed:
 *   mfm_plan_forward_train     = mfm_plan_forward(train=1, y=NULL, no outputs): x_hat / y_hat stay in the workspace
 *                                (mfm_plan_out_layout), the loss slots [1..4] are filled, slot [0] stays 0; `grads_to_zero`
 *                                (optional, the plan's flat layout) is cleared inside the first launch.
 *   mfm_plan_backward_weighted = backward of  w.disc * L_disc(y_hat, y) + sum_m w.gen_m * MSE_m + w.reg * reg  on the last
 *                                forward.  w.gen_* must equal the plan's lda_x* or all be 0 (MFM_ERR_UNSUPPORTED otherwise:
 *                                the caller falls back to mfm_plan_backward_ext); write_disc_loss != 0 adds L_disc into loss
 *                                slot [0] (the forward ran without labels).  `grads` is overwritten (cleared first unless
 *                                this step's forward_train already cleared it). */
typedef struct MfmLossWeights { float disc, gen_l, gen_a, gen_v, reg; int32_t write_disc_loss; } MfmLossWeights;
int mfm_plan_forward_train(MfmPlan* plan, const float* params, const float* x, uint64_t seed, void* workspace,
                           float* grads_to_zero, void* stream);
int mfm_plan_backward_weighted(MfmPlan* plan, const float* params, const float* x, const void* y,
                               const MfmLossWeights* w /*host*/, void* workspace, float* grads, void* stream);
/* byte offsets inside the workspace of x_hat_l / x_hat_a / x_hat_v [T,B,d_*] (out[0..2]; -1 on bf16-resident plans, which
 * do not keep them) and y_hat [B,output_dim] (out[3]) as the last forward left them; out[4..7] reserved (-1) */
int mfm_plan_out_layout(const MfmPlan* plan, int64_t* out /*[8]*/);

/* forward (train mode) + backward of the joint loss in one enqueue, no optimizer: the data-parallel step
 * all-reduces `grads` next and then calls mfm_adam_flat (factorized_amd/train.py).  Same launches as
 * mfm_plan_train_step minus Adam; the gradient buffer is cleared inside the first launch. */
int mfm_plan_grad_step(MfmPlan* plan, const float* params, float* grads, const float* x, const void* y,
                       uint64_t seed, void* workspace, float* losses, void* stream);

/* forward + backward + Adam in one enqueue (no host work in between). */
int mfm_plan_train_step(MfmPlan* plan, float* params, float* grads, float* adam_m, float* adam_v,
                        const float* x, const void* y, uint64_t seed, int32_t step, float lr,
                        float grad_scale, void* workspace, float* losses, void* stream);

/* forward (train mode) + backward of the STAGE loss (0 joint, 1 gen+reg, 2 disc+reg) + Adam over `spans`
 * (mfm_adam_flat_spans semantics) in one enqueue: one step of train_beta_vae's loop (reference mfm_mosi.py:255-285). */
int mfm_plan_train_step_staged(MfmPlan* plan, float* params, float* grads, float* adam_m, float* adam_v,
                               const float* x, const void* y, uint64_t seed, int32_t stage,
                               const MfmAdamSpan* spans /*host*/, int32_t nspans, float lr, float grad_scale,
                               void* workspace, float* losses, void* stream);

/* ---- per-plan switches (ABI 3).  Everything a plan decides is decided from its sizes; these are the switches a caller (or
 * a test) may set explicitly.  Unknown keys return MFM_ERR_ARG.
 *   "handover"            1 (default): B <= 32 plans of MFM_KL_EF run their input projections and weight gradients on role
 *                         workgroups INSIDE the encoder launches (in-launch hand-overs; the launch needs the GPU to itself).
 *                         0: separate launches, no workgroup ever waits for another one.  The host flips it to 0 when a
 *                         hand-over timed out (mfm_plan_status) or when several ranks share one device.
 *   "handover_timeout_us" how long a consumer spins before it gives up (default 50000).
 *   "grad_guard_offset"   element offset inside the gradient buffer of the guard word of mfm_adam_flat_guarded (default -1:
 *                         none; a wait that gives up then stores its NaN into grads[0]).  With an offset set, the plan's own
 *                         Adam launches are guarded, and every backward stores a NaN there while the status word is non-zero.
 *   "bf16_dot"            bf16 plans at batch sizes below the bf16 MFMA recurrences (B < 192): 1 = the one-row recurrences
 *                         that have the variant take their recurrent product on bf16 dot products (MfmSeqDesc::bf16_dot);
 *                         default 0 (measured: profiles/r04_bf16_onerow.txt; MFM_BF16_DOT=1 sets the default for new plans).
 *   "inject_fault"        fault injection for tests, one shot: 1 = one projection producer of the next forward does not raise
 *                         its flag; 2 = one BPTT workgroup of the next backward does not stamp its last gate gradients.
 *                         The waiting side must time out, set the status word and poison the guard. */
int mfm_plan_set_option(MfmPlan* plan, const char* key, int64_t value);
/* The tuning / test switches named MFM_* ("Kernel selection and its overrides") belong to the PLAN: its table is
 * filled from the environment when the plan is created and consulted -- never the environment -- by every launch the plan
 * issues.  This call changes one entry afterwards (value NULL removes it); switches that shape the workspace (MFM_BF16_STORE,
 * MFM_LATENT_*, ...) are read at creation only.  The granular entry points of this header, which have no plan, read the
 * environment at each call. */
int mfm_plan_set_option_str(MfmPlan* plan, const char* key, const char* value);
/* also answers two read-only keys: "proj_roles_active" / "dw_roles_active" = 1 when the plan's last forward / backward ran on
 * role workgroups */
int mfm_plan_get_option(const MfmPlan* plan, const char* key, int64_t* value);

/* Device-side state of a plan inside its workspace: out[0] byte offset of the plan's own loss slots [MFM_LOSS_SLOTS] floats
 * (used when `losses` is NULL), out[1] byte offset of the STATUS word (uint32, right behind them: 0 = fine; bit 0 a consumer of
 * projections gave up waiting, bit 1 a weight-gradient block did; sticky -- only the host clears it, mfm_plan_clear_status),
 * out[2] byte offset of the replay counter (uint64) and out[3] of the backward replay counter (uint32): device words a
 * captured hipGraph advances on every replay (the dropout streams and the hand-over epochs add them), out[4..7] reserved. */
int mfm_plan_state_layout(const MfmPlan* plan, int64_t* out /*[8]*/);
/* The same failure, visible WITHOUT a copy or a synchronisation: two uint32 words of host-coherent pinned memory owned by the
 * plan (allocated by mfm_plan_init_workspace; NULL before).  A consumer that gives up stores 1 into word [0] (projections) /
 * 2 into word [1] (weight gradients) with a system-scope store next to the device status word; an optimizer polls them for
 * free before every update (factorized_amd/optim.py).  mfm_plan_clear_status clears them too. */
int mfm_plan_host_status(MfmPlan* plan, uint32_t** out /*host pointer*/);
/* enqueue the clearing of the status word */
int mfm_plan_clear_status(MfmPlan* plan, void* workspace, void* stream);

/* Where the latent stack keeps its per-row record inside the workspace (tests and tuning aids: dropout masks and
 * pre-activation gradients can be read back from the workspace tensor).  out[0] byte offset of the activation
 * record [B, rec], out[1] byte offset of the gradient record, out[2] rec (floats per row); then record offsets
 * (floats): out[3..6] dropout scale (0 or 1/(1-p)) of z{l,a,v,y}_to_f*_fc1, out[7] of fy_to_y_fc1,
 * out[8..11] post-dropout activation of the same four layers, out[12] of fy_to_y_fc1, out[13..16] their widths
 * f{l,a,v,y}, out[17..20] f segments (outputs of *_fc2), out[21] y_hat, out[22] 1 if the row-per-workgroup kernels
 * run at this (T,B), out[23..26] mu segments z{l,a,v,y} (inputs of *_fc1), out[27..30] their widths, out[31] reserved. */
int mfm_plan_latent_layout(const MfmPlan* plan, int64_t* out /*[32]*/);

/* variants 1, 2: where the Memory Fusion Network keeps its [T*B, .] tensors inside the workspace (tests / tuning aids).
 * Byte offsets: out[0] cStar, out[1] h1 = drop(relu(att1_fc1)), out[2] its relu/dropout mask (0 | 1/(1-p)), out[3] attention,
 * out[4] attended, out[5] h2, out[6] its mask, out[7] cHat, out[8] final memory [B, memsize], out[9] the latent stack's y
 * input [B, nzy]; then sizes: out[10] width of cStar, out[11] NN1, out[12] NN2 hidden widths, out[13] memsize, out[14] nzy,
 * out[15] T*B. */
int mfm_plan_mfn_layout(const MfmPlan* plan, int64_t* out /*[16]*/);

/* Where LSTM `which` keeps its saved activations inside the workspace (tests / tuning aids).  which: 0 .. n_enc-1 the
 * encoders in plan order (l, a, v, early-fusion | l, a, v, MFN l, a, v), then the three decoders.  Byte offsets out[0] gates
 * [T,B,4,Hp], out[1] hs [T,B,Hp], out[2] cs [T,B,Hp] (always fp32); out[3] h, out[4] Hp; out[5] 1 when the plan is
 * bf16-RESIDENT: gates, hs, the decoders' dH and d x_hat are __bf16 buffers; out[6] 1 for a decoder; decoders: out[7] dH
 * [T,B,Hp], out[8] d x_hat [T*B, out[9]] (out[10] real columns); encoders: out[7] the fp32 copy of h_{T-1} [B,Hp] (-1: none);
 * out[11] 1 when the recurrences run on the bf16 MFMA kernels. */
int mfm_plan_seq_layout(const MfmPlan* plan, int32_t which, int64_t* out /*[12]*/);

/* Algorithmic work of one training step at this plan's (T,B):
 * 3 x forward FLOPs; activation+input bytes per sample plus 10 P 4 parameter/optimizer bytes. */
double mfm_plan_flops_per_step(const MfmPlan* plan);
double mfm_plan_bytes_per_step(const MfmPlan* plan);

/* Per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline object).
 * mask bit k enables kernel id k (ids 0..mfm_plan_num_kernels()-1, names from
 * mfm_plan_kernel_name); mfm_plan_collect_timing synchronises on the recorded events, returns
 * summed milliseconds and launch counts per kernel id and resets the recorder. */
int mfm_plan_set_timing(MfmPlan* plan, int mask);
/* bracket only every `every`-th step (default 1 = every step): an event bracket is two extra packets on the stream
 * (mfm_timing_bracket_overhead_ms), which a timed region should not pay on every step */
int mfm_plan_set_timing_every(MfmPlan* plan, int every);
int mfm_plan_num_kernels(void);
const char* mfm_plan_kernel_name(int kid);
int mfm_plan_collect_timing(MfmPlan* plan, double* sum_ms, int64_t* count);
/* median cost (ms) of an EMPTY event bracket on `stream`: what a bracket adds to the kernel it surrounds */
int mfm_timing_bracket_overhead_ms(void* stream, double* ms_out);
/* algorithmic FLOPs of ONE launch of kernel `kid` (matrix work only). */
double mfm_plan_kernel_flops(const MfmPlan* plan, int kid);

#ifdef __cplusplus
}
#endif
#endif /* MFM_HIP_H_ */
