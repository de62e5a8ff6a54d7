/*
 * lgd_hip.h — C ABI of liblgd_hip.so: the hand-written gfx950 (MI355X) kernels of the
 * LMD / LMD+ stage-2 denoising hot path.
 *
 * The reference (TonyLianLong/LLM-groundedDiffusion) is pure Python on PyTorch; it has no FFI of
 * its own.  Each entry point below therefore replaces a *PyTorch op sequence* of the reference and
 * cites it (file:line under the reference root).  The ctypes binding a maintainer would add is
 * llm-groundeddiffusion_amd/_lib.py.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the parameter is documented "host";
 *  - activations / weights are IEEE fp16 (`_Float16`), statistics / biases / losses fp32;
 *  - feature maps are channels-last: [B][H*W][C] row-major ("NHWC"); the UNet boundary tensors
 *    (latents in, noise prediction out) are NCHW fp32 exactly as the reference passes them;
 *  - no allocation, no synchronisation, no host<->device copy inside any call: every output and
 *    workspace is caller-allocated; kernels are enqueued on `stream` (a hipStream_t passed as
 *    void*) and the call returns immediately — all calls are hipGraph-capturable;
 *  - return value: 0 on success, negative LGD_ERR_* otherwise (the Python side raises RuntimeError,
 *    the error convention of the reference's plugin boundary: generate.py:391-396).
 */
#ifndef LGD_HIP_H
#define LGD_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LGD_ABI_VERSION 10
int lgd_abi_version(void);
/* Kernel-variant switches of the library (A/B timing and tests; the defaults are what the benchmark runs).  No
 * counterpart in the reference.  "attn32": self-attention forward without map capture — 0 = the 16x16x32 kernel,
 * 1 = (default) the 32x32x16 software-pipelined kernel where it applies (d + 2 <= 96, enough work to fill the chip),
 * 2 = that kernel for every problem size.  "attn_w4": the d = 40 kernel of round 4 (csrc/attn_w4.hip: 4-wave
 * workgroups, LDS-DMA K / V, transposing V reads) — 0 = never, 1 = (default) once a launch has >= 256 workgroups of
 * 256 queries and >= 256 keys, 2 = for every problem size; "attn_w4_pipe": 1 = (default) one wave per SIMD with the
 * in-wave software pipeline, 0 = two waves per SIMD.  "gn_fused": the largest map (pixels per image) lgd_groupnorm_f16
 * normalises in ONE launch (a workgroup holds its image x groups slab in registers); default 256 (16x16), 0 = always
 * the two-launch form.  "gn_slab": 1 = (default) lgd_groupnorm_bwd_f16 runs in one launch where a workgroup can hold its
 * (image, groups) slab of x and gy in registers (<= 96 KB: the 8x8 and 16x16 maps), 0 = two launches.  "ln_stream": 1 =
 * (default) the statistics-only form of lgd_layernorm_f16 (y = NULL) runs the streaming kernel (lane groups share a row),
 * 0 = the one-wave-per-row kernels.  "gn_apply_wgs" (tools): the number of workgroups per launch the GroupNorm apply passes
 * aim at, 64 .. 8192, default 1024.  Returns 0, or LGD_ERR_ARG for an unknown name. */
int lgd_set_option(const char* name, int value);

/* ---------------------------------------------------------------------------------------------
 * GEMM / implicit-GEMM convolution on MFMA (v_mfma_f32_16x16x32_f16).
 *   C[m][n] = epilogue( sum_k A(m,k) * W[n][k] )
 * Replaces: nn.Linear / nn.Conv2d calls of the UNet — attention_processor.py:338-363,426-453
 * (to_q/to_k/to_v/to_out), attention.py:286-289,333-335 (FeedForward/GEGLU), transformer_2d.py:
 * 283-291,319-325 (proj_in/proj_out), [ext diffusers 0.18.0] ResnetBlock2D conv1/conv2/
 * conv_shortcut, Downsample2D, Upsample2D (call sites unet_2d_blocks.py:186-197,315-326,360-362,
 * 577-588,621), and their input-gradient (dgrad) forms used by pipelines.py:56.
 *
 * A operand ("taps" = 1: plain rows; "taps" = 9: 3x3 gather, zero padding 1):
 *   A(m,k): k -> (tap, c) with c in [0, c0+c1); c < c0 reads source a0 else a1 (channel concat of
 *   two feature maps without materialising torch.cat, unet_2d_blocks.py:646-649);
 *   taps=9: m -> (b, oy, ox) over hout*wout; input pixel (oy*stride+ky-1, ox*stride+kx-1); with
 *   ups=1 the logical input is the nearest-2x upsampling of the stored hin*win map; with ups=2
 *   it is the zero-inserted map (data at even coordinates only: dgrad of a stride-2 conv).
 * Epilogue (in this order): +bias[n] +bias2[n]; GEGLU pairs (value, gate) column blocks of 16 and
 * emits value*gelu(gate) (N/2 output columns); *alpha; +res[m][n]; store fp16 (or fp32).
 * ------------------------------------------------------------------------------------------- */
#define LGD_EPI_GEGLU 1   /* weights/bias rows packed [16 value | 16 gate] blocks               */
#define LGD_EPI_OUT_F32 2 /* C is fp32                                                           */
#define LGD_EPI_RES_F32 4 /* res is fp32                                                         */
#define LGD_EPI_ROWNORM 8 /* LayerNorm of the A rows folded into the GEMM (ABI v8): the contraction runs on the RAW rows
                             x with W' = W * gamma (per input channel), and the epilogue computes, per row m and column n,
                             rstd[m] * (acc - mean[m] * colsum[n]) before bias / GEGLU / alpha / residual, where
                             colsum[n] = sum_k W'[n][k] and the bias holds b[n] + sum_k beta[k] W[n][k]:
                             LN(x) W^T + b  =  rstd (x W'^T - mean colsum) + b'   (attention.py:185,206,223 followed by
                             attention_processor.py to_q/k/v, attention.py:286-289 GEGLU).  Needs rowstat + colsum. */

typedef struct LgdGemmDesc {
  const void* a0;
  const void* a1;
  int64_t lda0, lda1; /* row (pixel) stride of each source, elements                       */
  int32_t c0, c1;     /* channels taken from a0 / a1; K = taps*(c0+c1)                      */
  int32_t taps;       /* 1 or 9                                                             */
  int32_t hin, win, hout, wout, stride, ups;
  const void* w;      /* [N][K] fp16                                                        */
  int64_t ldw;
  int32_t M, N, K;
  int32_t nb_o, nb_i; /* batch = nb_o*nb_i problems (e.g. image x head)                     */
  int64_t a_bs_o, a_bs_i, w_bs_o, w_bs_i, c_bs_o, c_bs_i, r_bs_o, r_bs_i;
  const float* bias;  /* [N] or NULL                                                        */
  const float* bias2; /* [N] or NULL (time-embedding projection of the current step)        */
  const void* res;    /* [M][ldr] or NULL                                                   */
  int64_t ldr;
  float alpha;
  int32_t epi;
  void* c;
  int64_t ldc;
  int32_t splits;     /* split-K factor (>=1); >1 needs ws                                  */
  float* ws;          /* fp32 [batch][splits][M][N]                                         */
  int32_t tile;       /* 0 auto; 1: 128x128, 2: 128x64, 3: 64x128, 4: 64x64, 5: 32x128, 6: 128x160, 7: 64x160
                         (register-staged main loop); +16 = same tile, LDS-DMA main loop (K % 64 == 0);
                         25: 256x320, 26: 256x128 (8 waves, LDS-DMA only);
                         33..42: 8-wave pipelined main loop, 3-6 LDS stages (K, c0, c1 % 64 == 0): 33: 256x160,
                         34: 256x128, 35: 256x64, 37: 128x160, 38: 128x128, 39: 128x64, 40: 64x160, 41: 64x128,
                         42: 64x64 (160-wide tiles: no GEGLU);
                         44 / 45 (ABI v8): two-stage rings, plain single-source contractions only (taps == 1,
                         c1 == 0): 44: 256x256 (one split, fp16 rows of whole 16-byte pieces, no fp32 / GEGLU
                         residual), 45: 128x128 with two workgroups per CU.  LGD_ERR_ARG where a code does not apply */
  int32_t* cnt;       /* split-K arrival counters (ABI v5), device int32[batches * tiles], ALL ZERO on entry, or NULL.
                         With counters the split-K combine happens inside the GEMM launch: every workgroup stores its
                         fp32 partial, publishes it (agent-scope release) and takes a ticket; the last arriver of an
                         output tile sums the `splits` partials in split order 0,1,2.. (bit-identical to the separate
                         reduce kernel), applies the epilogue and leaves the counter at zero again.  NULL = second
                         launch (splitk_reduce_kernel).  One counter buffer may serve every GEMM of a stream. */
  const float* rowstat; /* LGD_EPI_ROWNORM: fp32 [M][2] = (mean, rstd) of every A row (lgd_layernorm_f16 with y = NULL) */
  const float* colsum;  /* LGD_EPI_ROWNORM: fp32 [N] = row sums of the (gamma-scaled) weight matrix as stored          */
} LgdGemmDesc;

int lgd_gemm_f16(const LgdGemmDesc* desc /* host */, void* stream);

/* ---------------------------------------------------------------------------------------------
 * conv_in: latents NCHW fp32 (B,4,L,L) -> [B][L*L][Cout] fp16, 3x3 pad 1 (unet_2d_condition.py:860)
 * w: [Cout][3][3][Cin] fp16, bias fp32.
 * ------------------------------------------------------------------------------------------- */
int lgd_conv_in_f16(const float* x_nchw, const void* w, const float* bias, void* y, int B, int Cin,
                    int L, int Cout, void* stream);
/* conv_out: [B][L*L][Cin] fp16 (already GroupNorm+SiLU'd) -> NCHW fp32 (B,Cout,L,L), times
 * out_scale (unet_2d_condition.py:972-975).  w: [Cout][3][3][Cin] fp16.  Called with the
 * flipped/transposed conv_in weights it is also conv_in's input gradient (the latent gradient of
 * pipelines.py:56; out_scale then undoes the fp16 gradient scaling). */
int lgd_conv_out_f16(const void* x, const void* w, const float* bias, float* y_nchw, int B, int Cin,
                     int L, int Cout, float out_scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * GroupNorm (+SiLU) over channels-last maps, two launches: statistics then apply.
 * Replaces [ext] ResnetBlock2D norm1/norm2 + nonlinearity, transformer_2d.py:283 (eps 1e-6, no
 * SiLU) and unet_2d_condition.py:972-974.  x may be the channel concat of two maps (x0: c0
 * channels, x1: c1 channels).  part: fp32 workspace [B][nchunk][G][2]; stats: fp32 [B][G][2]
 * (mean, rstd) written by apply for the backward pass.
 * ------------------------------------------------------------------------------------------- */
int lgd_groupnorm_f16(const void* x0, const void* x1, int c0, int c1, int B, int HW, int G,
                      float eps, const float* gamma, const float* beta, int silu, void* y,
                      float* part, int nchunk, float* stats, void* stream);
/* backward of the above w.r.t. x: gy [B][HW][C] -> gx0 (c0 channels, row stride c0) and gx1.
 * accumulate!=0 adds into gx (gradient fan-in). */
int lgd_groupnorm_bwd_f16(const void* gy, const void* x0, const void* x1, int c0, int c1, int B,
                          int HW, int G, const float* gamma, const float* beta, int silu,
                          const float* stats, void* gx0, void* gx1, float* part, int nchunk,
                          int accumulate, void* stream);

/* LayerNorm over the last dim (attention.py:185,206,223; GatedSelfAttentionDense norm1/norm2
 * attention.py:35-36,50-51). rows x C, C % 8 == 0. y row stride ldy (lets the fuser write visual
 * tokens into the [S+30] concat buffer). stats [rows][2] (mean, rstd) optional. */
int lgd_layernorm_f16(const void* x, int64_t ldx, void* y, int64_t ldy, int rows, int C, float eps,
                      const float* gamma, const float* beta, float* stats, int rows_per_batch,
                      int64_t x_bs, int64_t y_bs, void* stream);
int lgd_layernorm_bwd_f16(const void* gy, int64_t ldgy, const void* x, int64_t ldx, void* gx,
                          int64_t ldgx, int rows, int C, const float* gamma, const float* stats,
                          int rows_per_batch, int64_t gy_bs, int64_t x_bs, int64_t gx_bs,
                          int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Scaled-dot-product attention, flash style (online softmax, K/V tiles staged in LDS, MFMA for
 * QK^T and PV).  Replaces F.scaled_dot_product_attention at attention_processor.py:355-357 and
 * the baddbmm/softmax/bmm path at :201-233,447 when no map is requested.
 *   q: [B][Sq][H*d] view with row stride ldq (so a fused QKV buffer can be passed), k/v likewise.
 *   o: [B][Sq][H*d] fp16.  lse (optional): fp32 [B][H][Sq] = log2-domain log-sum-exp, for backward.
 * ------------------------------------------------------------------------------------------- */
int lgd_attn_fwd_f16(const void* q, int64_t ldq, int64_t q_bs, const void* k, int64_t ldk,
                     int64_t k_bs, const void* v, int64_t ldv, int64_t v_bs, void* o, int64_t ldo,
                     int64_t o_bs, float* lse, int B, int H, int Sq, int Sk, int d, float scale,
                     void* stream);
/* backward: given q,k,v,o,do,lse -> dq,dk,dv (fp16, same views). delta: fp32 ws [B][H][Sq]. */
int lgd_attn_bwd_f16(const void* q, int64_t ldq, int64_t q_bs, const void* k, int64_t ldk,
                     int64_t k_bs, const void* v, int64_t ldv, int64_t v_bs, const void* o,
                     int64_t ldo, int64_t o_bs, const void* go, int64_t ldgo, int64_t go_bs,
                     const float* lse, float* delta, void* gq, int64_t ldgq, int64_t gq_bs, void* gk,
                     int64_t ldgk, int64_t gk_bs, void* gv, int64_t ldgv, int64_t gv_bs, int B, int H,
                     int Sq, int Sk, int d, float scale, void* stream);
/* The same with dK / dV computed for the first Sk_grad <= Sk keys only (ABI v10); dQ still sums over all Sk keys.
 * GLIGEN's gated self-attention (attention.py:43-53) attends over [visual tokens ; 30 grounding tokens] and keeps the
 * visual rows; the grounding rows of the concatenated input are constants of a run, so nothing reads the gradient of
 * their keys / values, and the key block that holds them would cost the dK/dV pass a whole extra round of workgroups
 * (4096 + 30 keys = 17 blocks of 256 per (image, head) on a grid that 16 fill exactly).  Rows >= Sk_grad of gk / gv are
 * not written. */
int lgd_attn_bwd_keys_f16(const void* q, int64_t ldq, int64_t q_bs, const void* k, int64_t ldk,
                          int64_t k_bs, const void* v, int64_t ldv, int64_t v_bs, const void* o,
                          int64_t ldo, int64_t o_bs, const void* go, int64_t ldgo, int64_t go_bs,
                          const float* lse, float* delta, void* gq, int64_t ldgq, int64_t gq_bs, void* gk,
                          int64_t ldgk, int64_t gk_bs, void* gv, int64_t ldgv, int64_t gv_bs, int B, int H,
                          int Sq, int Sk, int Sk_grad, int d, float scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Cross-attention over the 77 text tokens with probability-map capture — the hook of
 * attention_processor.py:426-480 (slow path).  Whole K/V of a head lives in LDS.
 *   probs (optional): fp32 [Bp][H][Sq][Tp] where, following :466-476,
 *     tok < 0  : all Sk columns are stored (Tp = Sk)
 *     tok >= 0 : only column `tok` (Tp = 1)                      (return_token_ca_only=int)
 *     cond_only: only batch items b >= B/2 are stored (Bp = B/2) (return_cond_ca_only)
 * ------------------------------------------------------------------------------------------- */
int lgd_cross_attn_fwd_f16(const void* q, int64_t ldq, int64_t q_bs, const void* k, int64_t ldk,
                           int64_t k_bs, const void* v, int64_t ldv, int64_t v_bs, void* o,
                           int64_t ldo, int64_t o_bs, float* probs, int tok, int cond_only, int B,
                           int H, int Sq, int Sk, int d, float scale, void* stream);
/* backward w.r.t. q only (text K/V are constants of the run): recomputes P; takes the upstream
 * gradient on the output (go, may be NULL) and on the probability map (gp fp32 [B][H][Sq][Sk], may
 * be NULL) — the map is an output with its own gradient (guidance.py:244-286 via pipelines.py:56). */
int lgd_cross_attn_bwd_f16(const void* q, int64_t ldq, int64_t q_bs, const void* k, int64_t ldk,
                           int64_t k_bs, const void* v, int64_t ldv, int64_t v_bs, const void* go,
                           int64_t ldgo, int64_t go_bs, const float* gp, void* gq, int64_t ldgq,
                           int64_t gq_bs, int B, int H, int Sq, int Sk, int d, float scale,
                           void* stream);

/* Causal self-attention (key j visible to query i iff j <= i), exact softmax: the CLIP text encoder's attention
 * ([ext] transformers 4.29.2 CLIPAttention with the causal mask of CLIPTextTransformer; called from
 * models/models.py:67-80 and pipelines.py:303-304 through `text_encoder(...)`).  Views as lgd_attn_fwd_f16. */
int lgd_attn_causal_fwd_f16(const void* q, int64_t ldq, int64_t q_bs, const void* k, int64_t ldk, int64_t k_bs,
                            const void* v, int64_t ldv, int64_t v_bs, void* o, int64_t ldo, int64_t o_bs, int B,
                            int H, int S, int d, float scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Elementwise pieces.
 * ------------------------------------------------------------------------------------------- */
/* GEGLU backward (attention.py:333-335): h = proj(x) packed [16 value|16 gate] blocks, width 2*n;
 * gy [rows][n] -> gh [rows][2n] in the same packed layout. */
int lgd_geglu_bwd_f16(const void* h, const void* gy, void* gh, int64_t rows, int n, void* stream);
/* GEGLU forward on a stored packed pre-activation h [rows][2n] -> y [rows][n] (grad-enabled pass). */
int lgd_geglu_fwd_f16(const void* h, void* y, int64_t rows, int n, void* stream);
/* y = x * sigmoid(1.702 x) (fp16): CLIP text encoder MLP activation ([ext] transformers CLIPMLP, "quick_gelu"). */
int lgd_quick_gelu_f16(const void* x, void* y, int64_t n, void* stream);
/* NCHW fp32 (B, C <= 8, HW) -> channels-last fp16 [B*HW][8]: channels 0..C-1 = fp16(x), channels C..2C-1 (when 2C <= 8)
 * = fp16(x - fp16(x)), the rest zero.  Input of the UNet's conv_in (unet_2d_condition.py:860, 4 -> 320 channels) when
 * it runs as an implicit GEMM with K = 9*8 and the filter duplicated over the remainder channels. */
int lgd_nchw_to_nhwc8_f16(const float* x, void* y, int B, int C, int HW, void* stream);
/* y = a + b (fp16), n elements (gradient fan-in / residual). */
int lgd_add_f16(const void* a, const void* b, void* y, int64_t n, void* stream);
/* y = alpha * x (fp16) */
int lgd_scale_f16(const void* x, void* y, float alpha, int64_t n, void* stream);
/* y = softmax(scale * x) over the last dim, rows x n fp16 (VAE decoder mid-block attention, [ext]
 * AutoencoderKL; pipelines.py:117-127 decode). */
int lgd_softmax_rows_f16(const void* x, void* y, int64_t rows, int n, float scale, void* stream);
/* sum over the 2x2 children of a nearest-2x upsampling: gy [B][2H*2W][C] -> gx [B][H*W][C] */
int lgd_upsample2x_bwd_f16(const void* gy, void* gx, int B, int H, int W, int C, void* stream);

/* Classifier-free guidance + DDIM (eta=0) step + frozen-mask blend + latent history in one pass
 * (pipelines.py:436-453; [ext] DDIMScheduler.step).  eps: NCHW fp32 (2B,C,L,L) = [uncond; cond].
 *   e = eu + gs*(ec-eu); epsilon or v prediction; x0 = (x - sqrt(1-a_t) e)/sqrt(a_t);
 *   x' = sqrt(a_p) x0 + sqrt(1-a_p) e;
 *   if step < frozen_steps: x' = frozen_ref[step+1]*mask + x'*(1-mask)   (mask [B][HW] fp32)
 *   hist[step+1] = x'  (save_all_latents) when hist != NULL.   x_out may alias x.
 * coef_table: device fp32 [T][4] = {a_t, a_prev, guidance_scale, v_prediction flag}.
 * dyn: device int32[2] = {step, frozen_steps} — read on the device so that one captured hipGraph
 * replays for every step and for both stages. */
int lgd_cfg_ddim_step_f32(const float* eps, const float* x, float* x_out, const float* coef_table,
                          const int32_t* dyn, const float* frozen_ref, const float* mask, float* hist,
                          int B, int C, int HW, void* stream);
/* The same fused step for LINEAR MULTISTEP samplers — [ext] diffusers DPMSolverMultistepScheduler (dpmsolver++,
 * order 2, midpoint; models/models.py:46-47 `use_dpm_multistep_scheduler`), whose update is linear in the latents, the
 * current data prediction x0 and the previous one:
 *   m = eu + gs*(ec-eu);  x0 = c0 x + c1 m;  x' = A x + B x0 + C x0_prev;  x0_prev <- x0;  blend / hist as above.
 * coef_table: device fp32 [T][8] = {c0, c1, A, B, C, guidance_scale, 0, 0} (host: scheduler.DPMSolverMultistepScheduler).
 * x0_prev: fp32 state buffer (B,C,L,L), read only when C != 0. */
int lgd_cfg_multistep_step_f32(const float* eps, const float* x, float* x_out, float* x0_prev,
                               const float* coef_table, const int32_t* dyn, const float* frozen_ref,
                               const float* mask, float* hist, int B, int C, int HW, void* stream);
/* Model-input scaling of sigma-space samplers — [ext] diffusers EulerDiscreteScheduler.scale_model_input, which the
 * SDXL-refiner pass applies before every UNet call (generation/sdxl_refinement.py:29 -> StableDiffusionXLImg2ImgPipeline):
 *   out[r][i] = x[i] * table[dyn[0] * row_stride + col]   for r < reps   (reps = 2: the CFG pair reads one latent).
 * The factor is read on the device, so the call sits inside a captured hipGraph that replays for every step. */
int lgd_scale_rows_f32(const float* x, float* out, const float* table, const int32_t* dyn, int row_stride, int col,
                       int64_t n, int reps, void* stream);
/* guidance latent update (pipelines.py:60-69): x -= active[i/per_sample] * coef_table[*step_idx][col] * g.
 * active (device fp32 per image, or NULL = all on) emulates the per-image `while` exit of
 * pipelines.py:30 when several layouts are guided in one batch. */
int lgd_axpy_f32(const float* g, float* x, const float* coef_table, const int32_t* step_idx, int col,
                 const float* active, int64_t per_sample, int64_t n, void* stream);
/* copy row `*idx` (device int32) of a [T][n] fp32 table into out[n] — per-step time-embedding
 * bias of every resnet without changing any kernel argument (graph-replay friendly). */
int lgd_select_row_f32(const float* table, const int32_t* idx, float* out, int n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * SAM mask refinement (models/sam.py:25-55 -> [ext] transformers SamModel): pieces of the ViT image encoder and the
 * mask decoder that are not GEMM / LayerNorm / attention calls above.
 * ------------------------------------------------------------------------------------------- */
#define LGD_ACT_GELU 1 /* exact (erf) GELU: SamMLPBlock of the image encoder, mask-decoder upscaling */
#define LGD_ACT_RELU 2 /* SamMLPBlock / SamFeedForward of the mask decoder */
/* y = act(x), fp16, n % 8 == 0. */
int lgd_act_f16(const void* x, void* y, int64_t n, int mode, void* stream);
/* Window partition + decomposed relative-position bias of SamVisionAttention, folded into the attention operands.
 * qkv [B*Hs*Ws][3*NH*d] fp16 (fused projection, raster token order), qkv_bias fp32 [3*NH*d] (value of the zero-padded
 * window positions), rel_h / rel_w fp32 [2*S-1][d] with S = window (window > 0: ceil(Hs/S) x ceil(Ws/S) windows,
 * padded) or S = Hs = Ws (window == 0: global attention).  Writes qa / ka / va [B*nwin*S*S][NH*DA] fp16,
 *   qa = [q | q.Rh[qy-j+S-1]/scale, j<S | q.Rw[qx-j+S-1]/scale, j<S | 0],  ka = [k | onehot(ky) | onehot(kx) | 0],
 *   va = [v | 0],   DA >= d + 2*S, DA % 8 == 0,
 * so that lgd_attn_fwd_f16(qa, ka, va, d = DA, scale) computes softmax(scale*q.k + rel_h + rel_w) v in columns 0..d-1
 * of every head. */
int lgd_sam_relpos_qkv_f16(const void* qkv, const float* qkv_bias, const float* rel_h, const float* rel_w, int B,
                           int Hs, int Ws, int window, int NH, int d, int DA, float scale, void* qa, void* ka,
                           void* va, void* stream);
/* Inverse gather (SamVisionLayer.window_unpartition): oa [B*nwin*S*S][NH*DA] in window order -> out [B*Hs*Ws][NH*d]
 * in raster order, padding positions and the DA-d extra columns dropped. */
int lgd_sam_window_merge_f16(const void* oa, void* out, int B, int Hs, int Ws, int window, int NH, int d, int DA,
                             void* stream);

/* ---------------------------------------------------------------------------------------------
 * Cross-attention energy of LMD / LMD+ and its gradient on the probability maps, one launch for
 * all (key, object, token, head) items: utils/guidance.py:91-148 (box loss, both branches: max-based fg/bg
 * top-k :131-145 and the ratio-based default :118-130 that generation/backward_guidance.py runs),
 * :150-242 (reference-attention L1 transfer), :244-286 (compute_ca_lossv3), times loss_scale
 * (pipelines.py:48).
 *   items: int32 [n_items][8] = {map_id, kind(0 topk, 1 ref, 2 ratio), token, mask_id, k_fg, k_bg, ref_id, image},
 *          sorted so that the items of one (map, image, token) column are adjacent;
 *   groups: int32 [n_groups][2] = {first item, item count} of each column — one workgroup per (group, head)
 *          sums the column's map gradients in a fixed order and stores them once (no atomics)
 *   coefs: fp32  [n_items][4] = {fg_coef, bg_coef, ref_coef, ratio_coef} (all normalisations folded in;
 *          kind 2: term = ratio_coef * (1 - sum(A*M)/sum(A))^2 per head, ABI v7)
 *   maps:  device array of n_maps pointers to fp32 [n_samples][H][HW][T]; gmaps likewise (pre-zeroed)
 *          or NULL; loss: fp32 [n_samples] (one value per image of the batch)
 *   map_hw: int32[n_maps]; masks: fp32 [n_masks][max_hw] (1 inside the box); refs: fp32
 *   [T][n_refs][H][max_hw] reference maps R_b (guidance.py:201); the slice of step dyn[0] (device
 *   int32) is used: refs + dyn[0]*refs_step_stride
 *   partial: fp32 [n_items*H] workspace; loss: fp32[1] = sum of all terms.
 *   grad_scale multiplies the map gradients only (static loss scaling for the fp16 backward pass;
 *   undone by the out_scale of the final conv_in dgrad).
 *   max_hw <= 4096 (ABI v9; 1024 before): guidance keys at the 64x64 level of a 512^2 SD 1.x network are legal, as
 *   utils/guidance.py accepts any `guidance_attn_keys`; larger maps return LGD_ERR_ARG.
 * ------------------------------------------------------------------------------------------- */
int lgd_ca_energy_f32(const float* const* maps, float* const* gmaps, const int32_t* map_hw,
                      const int32_t* items, const float* coefs, const float* masks, const float* refs,
                      int64_t refs_step_stride, const int32_t* dyn, const int32_t* groups, int n_groups,
                      int n_items, int n_samples, int H, int T, int max_hw, float grad_scale, float* partial,
                      float* loss, void* stream);

/* ---------------------------------------------------------------------------------------------
 * BoxDiff energy (ABI v9) and its gradient on the probability maps, one launch, one workgroup per image:
 * utils/boxdiff.py:20-101 (_compute_max_attention_per_index: x100 token soft-max of the layer- and head-averaged map
 * without its first and last token, reflect-padded 3x3 smoothing, inner- / outer-box top-k means, corner terms),
 * :104-118 (_compute_loss), :121-196 (compute_ca_loss_boxdiff), times amp_loss_scale (:224).  Replaces that Python loop
 * over objects x phrase tokens and the autograd graph through it (torch.cat / mean over the maps included).
 *   maps / gmaps: device arrays of n_maps pointers to fp32 [n_samples][H][side*side][T] (all maps of ONE resolution:
 *          generation/boxdiff.py:33-39 lists five 16x16 keys); gmaps pre-zeroed or NULL (value only)
 *   items: int32 [n_items][8] = {token (index in the 77-token prompt), mask_id, k_fg, k_bg, 0, 0, 0, 0}, the items of an
 *          image adjacent; k = (mask.sum() * P).long() as :80,:85 compute it; k = 0 drops that term (Python's
 *          max(0, nan) = 0, :107-109)
 *   groups: int32 [n_samples][2] = {first item, item count} per image; max_items = the largest count (sizes the LDS)
 *   masks: fp32 [n_masks][3][side*side]: row 0 the union of the object's boxes; row 1 corner_mask_x[side] |
 *          corner_mask_y[side] (:64-67); row 2 gt_proj_x[side] | gt_proj_y[side] (:90-91)
 *   smooth: 9 fp32 weights of GaussianSmoothing(kernel_size 3, sigma) (utils/attn.py:92-110) or NULL (no smoothing)
 *   loss: fp32 [n_samples] = loss_scale * energy; the map gradients carry loss_scale * grad_scale.
 * Returns LGD_ERR_UNSUPPORTED for side > 32, T > 128 or more items per image than fit the LDS (70 at 16x16).
 * ------------------------------------------------------------------------------------------- */
int lgd_boxdiff_energy_f32(const float* const* maps, float* const* gmaps, int n_maps, int side, const int32_t* items,
                           const float* masks, const float* smooth, const int32_t* groups, int n_samples,
                           int max_items, int H, int T, float loss_scale, float grad_scale, float* loss, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LGD_HIP_H */
