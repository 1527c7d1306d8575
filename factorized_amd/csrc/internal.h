// Internal (non-ABI) declarations shared between the .hip translation units of libmfm_hip.so.
#pragma once
#include <string.h>
#include "common.h"

namespace mfm {

// gemm.hip
#define MFM_GEMM_ZSPANS 4
#define MFM_GEMM_MAXP 56   // problems per launch (the descriptors travel in the kernel-argument segment, ~9.6 KB)
#define MFM_TN_MAXP 112    // ... of gemm_tn_launch, whose table is compact (gemm_tn.hip)
struct ZeroSpans { float* ptr[MFM_GEMM_ZSPANS]; int64_t n[MFM_GEMM_ZSPANS]; };   // spans (multiples of 4 floats, 16-byte aligned) a GEMM launch also clears
// optional per-problem output transform, applied to the finished element v of C (non-accumulating problems):
//   1  relu + dropout:  aux <- (v > 0) * scale,  v <- max(v, 0) * scale      scale = 0 | 1/(1-p) in train mode, else 1
//   2  tanh
//   3  v <- v * aux                                                            (backward of kind 1)
// aux is addressed like C (row * ldc + col).  The dropout stream is keyed by (seed, op_id, row, col).
struct GemmEpi { float* aux; float p; int kind; unsigned op_id; int pad_; };
struct GemmEpiSet { const GemmEpi* epi; int count; unsigned long long seed; int train;
                    const unsigned long long* tick; };   // tick: optional device word added to seed (the plan's replay counter)
struct MseEpi {          // squared-error epilogue of one product: target, d(output), loss slot, scales
  const float* x; int64_t ldx; float* dxhat; float* loss; float inv_count, grad_scale;
  int dxhat_bf16;        // dxhat is a __bf16 buffer (bf16-resident plans)
  int ld_dxhat;          // row stride of dxhat in elements (0: that of the product's C)
};
// precision: 0 = fp32 operands on v_mfma_f32_16x16x4_f32, 1 = operands rounded to bf16 on the way into LDS,
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation (gemm_bf16.hip)
// internal use of MfmGemmDesc::reserved_ (8 bytes, zero for every caller of the C ABI): TN products on gemm_tn_kernel can
// also add the column sums of their A operand (sum over the rows k of A[k][m], times alpha) into a vector -- the bias
// gradient that belongs to a weight gradient dW = G^T X -- from the slices they have in LDS anyway
static inline void gemm_set_colsum(MfmGemmDesc& d, float* p) { memcpy(d.reserved_, &p, sizeof(p)); }
static inline float* gemm_get_colsum_host(const MfmGemmDesc& d) { float* p; memcpy(&p, d.reserved_, sizeof(p)); return p; }
int gemm_group_launch(const MfmGemmDesc* descs, int count, hipStream_t stream, const ZeroSpans* zs = nullptr,
                      const MseEpi* mse = nullptr, int mse_count = 0, int precision = 0, const GemmEpiSet* epis = nullptr);
int device_cus();

// gemm_tn.hip -- the accumulating weight-gradient products C += A^T B at small row counts: (tile, 160-row chunk) workgroups
// that request their whole operand slices at once.  descs as for gemm_group_launch (count <= MFM_TN_MAXP).
// c_is_zero: the outputs of non-accumulating problems are known to hold zeros (the step's gradient buffer), so adding is storing
bool gemm_tn_supported(const MfmGemmDesc* descs, int count, int max_rows, bool c_is_zero);
int gemm_tn_launch(const MfmGemmDesc* descs, int count, int max_rows, bool c_is_zero, hipStream_t stream);

// lin_rows.hip -- row-block linear layers C = epi(A W^T + bias) for few rows (the MFN attention block at small T*B): every
// workgroup requests its whole 16 x k / 32 x k operand slices at once instead of walking K through a load ring.
// kind / aux / p / op_id as GemmEpi (kinds 0-3).
#define MFM_LINROWS_MAX 3
struct LinRowsItem {
  const float* a; int lda; const float* w; int ldw; const float* bias; float* c; int ldc; int n, k;
  int kind; float* aux; float p; unsigned op_id;
  int trans;          // 0: W[n, k] (forward), 1: Wt[k, n] -- C = A Wt, the input-gradient products of the backward
  int accumulate;     // atomicAdd into C (kind 0 only)
};
bool lin_rows_supported(const LinRowsItem* items, int count, int M);
int lin_rows_launch(const LinRowsItem* items, int count, int M, int train, unsigned long long seed, hipStream_t stream,
                    const unsigned long long* tick = nullptr);

// gemm_panel.hip -- row-panel GEMM for the large-batch input projections: the A rows stay in LDS, every column group
// (weight block [n_valid, k_len] with row stride ldw, consuming panel columns [k_off, k_off + k_len)) is walked by the
// same workgroup.  C columns [n_valid, n) are written as zeros (pad units).
#define MFM_PANEL_MAXG 28
struct PanelGroup {
  const float* w; const float* bias; const float* bias2; float* c;
  int64_t ldw, ldc;
  // n = nseg * seg output columns: column j belongs to segment j / seg (an LSTM gate) at position u = j % seg; it is a
  // real unit when u < seg_valid, and then weight row / bias element (j / seg) * seg_valid + u produces it; pad
  // columns are written as zeros.  (One LSTM = one group: nseg 4, seg Hp, seg_valid h.)
  int n, seg, seg_valid, k_off, k_len;
  int c_bf16, pad_;       // c is a __bf16 buffer (bf16-resident x-projection of a bf16 plan); ldc counts its elements
};
struct PanelLaunch {
  const float* a; int64_t lda; int M, K;
  PanelGroup g[MFM_PANEL_MAXG]; int ngroups;
  float* zero_ptr[MFM_GEMM_ZSPANS]; int64_t zero_n[MFM_GEMM_ZSPANS];
};
// force = false: the launcher declines (MFM_ERR_UNSUPPORTED, no error text) when its cost model says the tiled kernel is faster
int gemm_panel_launch(PanelLaunch& L, const ZeroSpans* zs, int precision, bool force, hipStream_t stream);
bool gemm_panel_pays(const PanelLaunch& L, int precision, bool force);

// proj_bf16.hip -- the same projections for a bf16-RESIDENT plan: weights packed once per step into a bf16 tile image
// (proj_bf16_pack_launch), tiles by LDS-DMA, bf16 results by 8-byte stores, the padded bf16 image of x as a side output
struct ProjPlan { int ntiles, nbias, BM, S, ns; size_t lds; int tile0[MFM_PANEL_MAXG], kt0[MFM_PANEL_MAXG], nkt[MFM_PANEL_MAXG], nchunks[MFM_PANEL_MAXG], bias_off[MFM_PANEL_MAXG]; };
int proj_bf16_plan(const PanelLaunch& L, ProjPlan* out);      // 1: supported (only the groups' shapes, M and K are read)
// scratch: wimg = ntiles * 8192 bytes (16-byte aligned), bimg = nbias floats
int proj_bf16_pack_launch(const PanelLaunch& L, const ProjPlan& P, void* wimg, float* bimg, hipStream_t stream);
int proj_bf16_launch(const PanelLaunch& L, const ProjPlan& P, const void* wimg, const float* bimg, void* x16, int x16_ld,
                     const int* xsrc0, const int* xn, const int* xdst0, const ZeroSpans* zs, hipStream_t stream);

// dec_fc1.hip -- decoder fc1 forward + squared-error loss + backward to the hidden states, one launch (fp32)
struct DecFc1Item {
  const float* hs; const float* w; const float* bias; const float* x;   // H [rows, Hp], Wfc [d, h], b [d], target columns (row stride ldx)
  float* xhat; float* dxhat; float* dhs; float* loss;                   // [rows, d] (optional), [rows, d] (optional), [rows, Hp], slot
  int64_t ldx; int d, h, Hp; float inv_count, grad_scale;
  int tile_begin, col_groups, frags_per_group;                          // filled by dec_fc1_launch
};
struct DecFc1Launch { DecFc1Item it[3]; int n_items, rows, with_bwd, bf16; };   // bf16: operands rounded to bf16 (RNE) first
int dec_fc1_launch(DecFc1Launch& L, bool dhs_zeroed, hipStream_t stream);

// dec_fc1_large.hip -- bf16-resident plans, large T*B: decoder fc1 + squared error + d x_hat + dH in one launch of
// persistent workgroups (the W image stays in LDS; H in, d x_hat and dH out are bf16 buffers)
struct DecFc1LargeItem {
  const void* hs; const float* w; const float* bias; const float* x;   // H [rows, Hp] bf16, Wfc [d, h] fp32, b [d], target columns (row stride ldx)
  void* dxhat; void* dhs; float* loss;                                   // [rows, ld_dxhat] bf16, [rows, Hp] bf16, loss slot
  int64_t ldx; int d, h, Hp, ld_dxhat; float inv_count, grad_scale;
  int wg_begin, wg_count;                                                // filled by dec_fc1_large_launch
  void* wimg;     // optional scratch, dec_fc1_large_wimg_bytes(): the bf16 image of Wfc, packed once per launch instead of once per workgroup
};
struct DecFc1LargeLaunch { DecFc1LargeItem it[3]; int n_items, rows, dbg, packed; };   // dbg: tuning aid (MFM_FC1_LARGE_DBG); packed: the weight images are already built
bool dec_fc1_large_uses_wimg(const DecFc1LargeLaunch& L);
int dec_fc1_large_supported(const DecFc1LargeItem& I);
size_t dec_fc1_large_wimg_bytes(int d);
int dec_fc1_large_launch(DecFc1LargeLaunch& L, hipStream_t stream);

// dw_bf16.hip -- bf16-RESIDENT plans: every sum over the T*B rows that feeds an LSTM's (or a decoder fc1's) weight gradient
// as one product C[M, N] += A^T [seg0 | seg1] per item, operands streamed by LDS-DMA and read with transposing LDS reads
#define MFM_DWB_MAXI 12
#define MFM_DWB_MAXOUT 4
struct DwbSeg { const __bf16* p; int ld, ncols, col0, shift, rows, pad_; };   // columns [col0, col0 + ncols) of a [rows, ld] bf16 buffer; chunk row r reads row r - shift
struct DwbOut { int n0, nvalid; float* c; float* c2; int ldc, pad_; };          // columns [n0, n0 + nvalid) of [seg0 | seg1] -> C[:, 0 .. nvalid) (row stride ldc)
struct DwbItem {
  const __bf16* a; int lda, M;          // A [rows, lda] bf16, columns [0, M) walked
  int Hp, h;                            // A column m = g * Hp + u is a real unit when u < h, and then row g * h + u of C
  int nseg, nout;
  DwbSeg seg[2]; DwbOut out[MFM_DWB_MAXOUT];
  float* cb; float* cb2;                // optional: column sums of A (bias gradients), indexed like the rows of C
  int tile_begin, m_tiles, splits, rows_per_split, stages, mt;       // filled by dw_bf16_launch (mt: A columns per workgroup tile)
  int64_t slab_off; int npad, red_begin;    // slab form (DwbLaunch::slabs): first float of this item's partial tiles, their row length, first row of the reduce launch
  // column parts (round 5, filled by dw_bf16_launch): an item whose right-hand side is split over `parts` consecutive entries, each
  // owning a column range of [seg0 | seg1]; the entries share tile_begin / m_tiles / splits and their workgroups are dealt out
  // row range by row range (all parts and M-tiles of one row range are neighbours on one XCD).  part > 0: tile_begin = INT_MAX.
  int parts, part;
  int kpc, pad2_;                       // 32-row k-blocks per chunk (filled by dw_bf16_launch: narrow items take 64 / 128 rows per chunk)
};
struct DwbLaunch { DwbItem it[MFM_DWB_MAXI]; int n_items, rows; const void* zeros; int debug_no_epilogue; int f32;
                   // optional scratch (dw_bf16_scratch_floats()): every workgroup leaves its partial tile there with plain stores and a
                   // second launch sums the row ranges of each M-tile and adds the result once -- instead of ~7 M atomicAdds into
                   // the same few hundred KB (round 4); null: the atomics
                   float* slabs; int64_t slab_floats; int red_rows, pad_; };   // f32: the buffers hold fp32 (round 3; element counts / strides then count floats)   // zeros: >= 16 bytes of device zeros (null: the launcher's own)
int dw_bf16_supported(const DwbItem& I, int f32 = 0);
int dw_bf16_launch(DwbLaunch& L, hipStream_t stream);
int64_t dw_bf16_scratch_floats(int64_t rows);      // upper bound of what a launch over `rows` rows needs in DwbLaunch::slabs
// x [rows, D] fp32 -> bf16 [rows, ldo] with up to three column ranges moved to 16-aligned positions (pad columns zero)
int x_to_bf16_launch(const float* x, void* out, int64_t rows, int D, int ldo, const int* src0, const int* n, const int* dst0,
                     hipStream_t stream);

// elementwise.hip
struct MseItem {
  const float* xhat; const float* x; float* dxhat; float* loss_slot;
  int64_t ldx, rows; int d; float inv_count, grad_scale; int block_begin;
};
int mse_group_launch(const MseItem* items, int count, hipStream_t stream);
// guard: optional device float; the update is skipped unless it is exactly 0 (include/mfm_hip.h, mfm_adam_flat_guarded)
int adam_launch(float* p, const float* g, float* m, float* v, int64_t n, int step, float lr, float beta1,
                float beta2, float eps, float grad_scale, hipStream_t stream, const float* guard = nullptr);
int adam_spans_launch(float* p, const float* g, float* m, float* v, const MfmAdamSpan* spans, int nspans, float lr,
                      float beta1, float beta2, float eps, float grad_scale, hipStream_t stream, const float* guard = nullptr);
int fill_launch(float* p, int64_t n, float val, hipStream_t stream);

// mfn_att.hip -- row-wise glue of the MFN attention block (everything between its GEMMs)
struct MfnCs { const float* cs[3]; float* dcx[3]; int h[3]; int T, B; };     // the three MFN LSTMs' cell states [T,B,Hp]
int mfn_cstar_launch(const MfnCs& c, float* cstar, hipStream_t stream);
int mfn_dcs_scatter_launch(const MfnCs& c, const float* dcs, hipStream_t stream);
int mfn_softmax_fwd_launch(float* att, const float* cstar, float* attended, int64_t rows, int n, hipStream_t stream);
int mfn_softmax_bwd_launch(const float* datt, const float* att, const float* cstar, float* dlog, float* dcs, int64_t rows,
                           int n, hipStream_t stream);
// mfn_mem.hip -- the heads on mfn_last = [h_l, h_a, h_v](T-1) | mem_T (mu_y and, variant 1, logvar_y) folded into the
// memory recurrence launches: forward as the kernel's tail (mem_T is in its LDS), backward as its head (d mem_T and d h_T
// from d [mu_y | logvar_y]); two ~6 us GEMM launches less per step.  The heads' weight gradients stay in the tail GEMM.
struct MfnHeadsDev {
  int on, tot, nheads, zy, nzy;
  const float* seg[3]; int seg_ld[3], seg_n[3];     // last hidden state rows [B, ld] of the three MFN LSTMs
  const float* w[2]; const float* b[2];             // head weights [zy, tot + M], biases [zy]
  float* zyin;                                      // forward out [B, nzy]: head hd at columns hd * zy
  const float* dz;                                  // backward in [B, nzy]
  float* d_hT;                                      // backward out [B, tot]
};
int mfn_mem_fwd_launch(const MfmMemDesc* desc, const MfnHeadsDev* heads, hipStream_t stream);
int mfn_mem_bwd_launch(const MfmMemDesc* desc, const MfnHeadsDev* heads, hipStream_t stream);
// mmd.hip -- strided form of mfm_mmd_fwd_bwd: z / dz are column blocks of wider row-major buffers
int mmd_launch(const float* z, int64_t ldz, const float* g, int64_t ldg, int B, int dim, float* loss, float* dz, int64_t lddz,
               float dz_scale, hipStream_t stream);
// up to 4 terms that share B, the row strides and the loss slot in one launch
struct MmdItem { const float* z; const float* g; float* dz; int dim; };
// scratch (optional, mmd_scratch_floats(B, count) floats, 16-byte aligned): large batches run as Gram-matrix GEMMs (mmd.hip)
int mmd_group_launch(const MmdItem* items, int count, int64_t ldz, int64_t ldg, int64_t lddz, int B, float* loss,
                     float dz_scale, hipStream_t stream, float* scratch = nullptr);
int64_t mmd_scratch_floats(int B, int count);

// latent.hip -- the fused "latent stack": encoder fc1 heads, mu/logvar heads, z->f MLPs,
// classifier, KLD and discriminative loss, interpreted from a small op table.
#define MFM_LAT_MAXOPS 24
#define MFM_LAT_MAXSTAGES 8
#define MFM_LAT_ROW_THREADS 1024
struct LatOp {
  int in_off, out_off, K, N;   // record offsets (floats), fan-in, fan-out
  int64_t w_off, b_off;        // element offsets of weight [N,K] / bias [N] in the flat param buffer
  int relu;                    // relu on the output
  int mask_off;                // record offset of the dropout mask (scale) segment, -1 = no dropout
  float drop_p;
  int stage;
  int pfx_n, pfx_k;             // exclusive prefix sums of N / K over the ops of the same stage (host-filled)
  int chain;                    // modality chain of the layer: 0 l, 1 a, 2 v, 3 y (incl. the classifier)
};
struct WtImgItem;
struct LatentDev {
  // Op table in DEVICE memory (uploaded once by mfm_plan_init_workspace).  It must not live in the
  // kernel-argument segment: the kernels index it with a per-lane op id, and a divergent index
  // into kernarg makes the compiler copy the whole struct to scratch in every thread.
  const LatOp* ops;
  int nops, nstages;
  int stage_begin[MFM_LAT_MAXSTAGES + 1];
  int rec_size;                        // floats per batch row, multiple of 4
  int wpanel;                          // floats of LDS reserved for one stage's weights (0 = read from L2)
  int64_t span_off[MFM_LAT_MAXSTAGES]; // first parameter element of each stage's contiguous tensor span
  int span_len[MFM_LAT_MAXSTAGES];     // span length in floats (multiple of 4)
  // inputs: last hidden state of the 4 encoders (l, a, v, fused)
  const float* enc_h[4]; int64_t enc_ld[4]; int enc_n[4]; int in_off[4];
  // latent segments (l, a, v, y)
  int mu_off[4], lv_off[4], z_n[4];
  int f_off[4], f_n[4];                // post-MLP features (l, a, v, y)
  int yhat_off, od;
  // decoder initial inputs [fy | f_m] (l, a, v) and their gradients
  float* dec_init[3]; const float* d_dec_init[3]; int64_t dec_ld[3];
  // gradient wrt the encoders' last hidden state
  float* dh_last[4]; int64_t dh_ld[4];
  float* rec;                          // [B, rec_size] saved record
  float* yhat_out;                     // optional [B, od]
  const void* y; int loss_kind;
  const float* d_yhat_ext;             // optional upstream gradient wrt y_hat [B, od] (module path)
  const float* reg_w_ptr;              // optional device scalar: upstream gradient wrt the KLD sum
  float* losses;
  float* grd_out;                      // [B, rec_size] gradient record written by the backward
  float seed_w; const float* seed_w_ptr;   // weight of grd_seed (device scalar when set: the caller's upstream gradient wrt the regulariser)
  const float* grd_seed;               // optional [B, rec_size]: gradients injected into the record before the backward
                                       // walks the stages (the MMD regulariser's d reg / d z of the non-KL MFM)
  unsigned long long* dbg;             // optional: block 0 / thread 0 writes s_memtime at phase marks
  int B, rows_per_wg, rows_fwd, train, has_logvar;     // rows per workgroup of the staged kernels: backward / forward
  int skip_bias;                       // staged backward: leave the bias gradients to the weight-gradient launch (gemm_tn column sums)
  int mfma;                            // staged kernels: the layers' products on v_mfma_f32_16x16x4_f32 (rows per workgroup <= 16, every K % 4 == 0)
  int row_path;                        // 1: one batch row per workgroup, weights read straight from L2 (latent.hip)
  // row path: per-thread work items of every stage, precomputed by the host ([nstages][MFM_LAT_ROW_THREADS] int4,
  // encoding in latent.hip) and the number of threads that have an item in each stage
  const int* items_fwd; const int* items_bwd;
  int nitems_fwd[MFM_LAT_MAXSTAGES], nitems_bwd[MFM_LAT_MAXSTAGES];
  // row path, small batches: the four modality chains of a row (l, a, v, y: independent inside the stack) run as
  // nch = 4 workgroups on 4 CUs (grid = B * nch, item tables [chain][stage][thread]); the record is laid out chain by
  // chain, chain c owns floats [ch_lo[c], ch_hi[c]) of it.  nch == 1: one workgroup per row, tables at index 0.
  int nch, ch_lo[4], ch_hi[4];
  int pre;                             // chain workgroups of 512 threads that request every stage's weights up front (latent.hip)
  int row_threads;                     // threads per row-path workgroup: 512 when no chain stage has more items, else 1024
  int nitems_fwd_c[4][MFM_LAT_MAXSTAGES], nitems_bwd_c[4][MFM_LAT_MAXSTAGES];
  uint64_t seed;
  const unsigned long long* tick;      // optional device word added to `seed` when the kernel runs (the plan's replay counter:
                                       // a captured hipGraph draws new masks on every replay)
  float reg_w, disc_w, gen_w;
  float* disc_loss_out;   // backward, optional: the discriminative loss value (L1 / CE mean) is ADDED here (the loss-weighted
                          // backward of the module path: the forward ran without labels, its slot 0 is still zero)
  int grd_agent;          // grd_out leaves with agent-scope stores (read inside the same launch, dw_role_dev.h)
  int bwd_split;          // backward, 1: the stages from tail_from up, the discriminative and KLD seeds already ran (HEAD blocks of the
                          // decoder BPTT launch, latent_row_dev.h mode 3) and left their gradient record in grd_seed (seed_w = 1):
                          // this launch adds the decoders' seeds and walks the stages below tail_from
  int tail_from;          // first stage nothing of the decoders waits for (behind the z -> f MLPs): the classifier, the logvar heads,
                          // the losses -- the fold launch may leave them to tail blocks of the decoder launch (latent_row_dev.h, mode 3)
  int64_t n_params;       // floats in the parameter buffer (row path: the weight requests are buffer loads that drop what lies outside)
  int bias_tab, bias_n;   // row path: the last stage slot of the backward item table lists every thread's bias-gradient element
                          // (bias_n: the largest number of entries a workgroup kind has; it must not exceed the block size)
};
int latent_fwd_launch(const LatentDev& L, const float* params, hipStream_t stream);
int latent_fwd_tail_launch(const LatentDev& L, const float* params, hipStream_t stream);      // latent.hip: mode-3 tails alone
// lstm_seq_small.hip / lstm_seq.hip -- the decoder recurrences with the latent forward chains' tails on idle CUs
bool seq_small_dectail_supported(int T, int B, const int* dec_h, const LatentDev& LD);
int seq_dec_bwd_head_launch(const MfmSeqDesc* descs, int count, int T, int B, const float* const* wt_imgs, const LatentDev& lat,
                            const float* params, float* grads, hipStream_t stream);
int seq_dec_tail_launch(const MfmSeqDesc* descs, int count, int T, int B, const float* const* wf_imgs, const LatentDev& lat,
                        const float* params, hipStream_t stream);
// lstm_seq.hip / lstm_seq_small.hip -- the encoder recurrences of MFM_KL_EF with their rows' latent chains folded in
int seq_fold_launch(const MfmSeqDesc* descs, int count, int T, int B, bool bwd, const LatentDev& lat, const float* params,
                    float* grads, hipStream_t stream, const float* const* wt_imgs = nullptr, const struct WtImgItem* img_items = nullptr,
                    int n_img_items = 0, bool* img_written = nullptr);
int seq_fwd_img_launch(const MfmSeqDesc* descs, int count, int T, int B, const struct WtImgItem* items, int n_items, bool* written,
                       hipStream_t stream);
int seq_bwd_img_launch(const MfmSeqDesc* descs, int count, int T, int B, const float* const* wt_imgs, hipStream_t stream);
int seq_fwd_wf_launch(const MfmSeqDesc* descs, int count, int T, int B, const float* const* wf_imgs, hipStream_t stream);
int latent_bwd_launch(const LatentDev& L, const float* params, float* grads, hipStream_t stream);

}  // namespace mfm
