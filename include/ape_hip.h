/*
 * ape_hip.h -- C-ABI of libape_hip.so: the MI355X (gfx950) kernels behind the APE-L_D forward pass.
 *
 * This is the drop-in boundary for the hot path named in BASELINE.json.
 * Every entry point takes plain device pointers + sizes + a hipStream_t (passed as void*), never
 * allocates, never synchronises, never takes ownership.  Return value: 0 = launched, negative =
 * argument/launch error with a message available from ape_hip_last_error() (thread local).
 *
 * dtype codes: APE_DT_F32 = 0, APE_DT_BF16 = 1 (raw bfloat16 bits, uint16_t).
 * All tensors are row-major with an explicit leading dimension in ELEMENTS.
 *
 * The reference interface each entry point replaces is cited next to it (paths relative to the
 * reference repository shenyunhang/APE).
 */
#ifndef APE_HIP_H
#define APE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APE_DT_F32 0
#define APE_DT_BF16 1
#define APE_DT_F16 2 /* IEEE half storage, fp32 arithmetic: accepted by ape_hip_ms_deform_attn_forward (the reference evaluates in
                      * fp16, tools/train_net.py:642, and its CUDA op dispatches half, ms_deform_attn_cuda.cu:65); the model-level
                      * kernels store bf16 -- an fp16 module casts at its edge (ape_amd/layers/multi_scale_deform_attn.py) */

#define APE_ACT_NONE 0
#define APE_ACT_RELU 1
#define APE_ACT_GELU 2   /* exact erf GELU (nn.GELU default) */
#define APE_ACT_SWIGLU 3 /* interleaved (gate,up) columns -> silu(gate)*up, N/2 outputs */
#define APE_ACT_SILU 4

#define APE_MASK_NONE 0
#define APE_MASK_ZERO_INPUT 1  /* masked rows behave as if the A row were zero (out = bias) */
#define APE_MASK_ZERO_OUTPUT 2 /* masked rows are written as zero (value.masked_fill)      */

const char* ape_hip_last_error(void);
int ape_hip_abi_version(void);
int ape_hip_sizeof_args(int which); /* sizeof(ApeGemmArgs / ApeLayerNormArgs / ApeGroupNormArgs) for which = 0 / 1 / 2 */

/* ---------------------------------------------------------------------------------------------
 * GEMM:  C[M,N] = epi(alpha * A[M,K] . W[N,K]^T)     (nn.Linear weight layout, K contiguous)
 * Replaces F.linear / nn.Linear / 1x1 Conv2d / PatchEmbed / ConvTranspose2d(k2,s2) contractions:
 *   ape/modeling/backbone/vit_eva_clip.py:125-132 (SwiGLU), :225-232 (q/k/v), :264-268 (proj),
 *   ape/modeling/backbone/utils_eva02.py:212-216 (PatchEmbed), vit_eva_clip.py:806-842 (FPN convs),
 *   ape/layers/multi_scale_deform_attn.py:268-277,353 (value/offset/weight/output proj),
 *   ape/layers/fuse_helper.py:70-73 (VL projections), ape/layers/vision_language_align.py:36-49.
 * epilogue order: x = alpha*acc ; rowmask(ZERO_INPUT) ; + bias[n] ; RoPE (pairs) ; act ;
 *                 clamp(+-clamp) ; + residual[m,n] ; rowmask(ZERO_OUTPUT) ; store (out_dt).
 * bf16 inputs: K % 8 == 0, lda/ldw % 8 == 0, 16-byte aligned A/W.  trans_out writes C^T[N][M]
 * (ldc = leading dimension of C^T) and supports bias + activation only.
 * ------------------------------------------------------------------------------------------- */
typedef struct ApeGemmArgs {
  const void* A;          /* [M, lda] in_dt                               */
  const void* W;          /* [N, ldw] in_dt                               */
  void* C;                /* [M, ldc] (or [N, ldc] if trans_out) out_dt   */
  const float* bias;      /* [N] fp32 or NULL                             */
  const void* residual;   /* [M, ldr] res_dt or NULL                      */
  const uint8_t* rowmask; /* [M] or NULL                                  */
  const float* rope_cos;  /* [rope_rows, rope_hd] fp32 or NULL            */
  const float* rope_sin;
  int32_t M, N, K;
  int32_t lda, ldw, ldc, ldr;
  int32_t in_dt, out_dt, res_dt;
  int32_t act;
  int32_t mask_mode;
  int32_t trans_out;
  int32_t rope_rows, rope_hd, rope_cols; /* rotate columns n < rope_cols; table row = m % rope_rows */
  int32_t vec_ok;                        /* filled by the launcher */
  float alpha;
  float clamp; /* <= 0: no clamp */
  /* split-K (bf16, K % 32 == 0, no trans_out): `splitk` > 1 slices K over blockIdx.y; fp32 partial tiles go to
   * `workspace` ([splitk, M, N] floats, caller-owned) and a second kernel reduces them and applies the epilogue.
   * For launches with too few 128x128 tiles to keep the memory system busy (M = 900 decoder GEMMs, K = T). */
  int32_t splitk;
  int32_t tile64; /* 1: 64x64 block tiles (4x more blocks; for launches with few output tiles), needs K % 32 == 0 */
  float* workspace;
  /* LayerNorm folded into the GEMM that consumes it (the SwiGLU sub-LN, vit_eva_clip.py:129-131): with W' = W diag(g),
   * LN(h) W^T = rstd_m (h W'^T)[m,n] - rstd_m mean_m (sum_k W'[n,k]) + (W b_ln)[n].  The epilogue applies, right after
   * alpha:  acc = acc * rowscale[m] + rowshift[m] * colvec[n]   (rowscale = rstd, rowshift = -rstd*mean from
   * ape_hip_row_stats, colvec = row sums of W'; the constant term travels in `bias`).  All three NULL = off. */
  const float* rowscale;
  const float* rowshift;
  const float* colvec;
  /* optional packed form of the RoPE tables: [rope_rows, rope_hd / 2] (cos, sin) float pairs, one per rotate_half pair (2i, 2i+1)
   * -- valid when both columns of a pair share an angle (cos[m, 2i] == cos[m, 2i+1], the reference's VisionRotaryEmbeddingFast
   * repeats every frequency twice, vit_eva_clip.py:179-216).  Same arithmetic, half the table bytes; the 256-row tile kernel
   * stages its rows through LDS with the LDS-DMA of the last K tile instead of re-reading them from L2 per accumulator row
   * (measured on the q|k projection: the table / bias loads were 11.7 of the launch's 46.9 us).  NULL = cos / sin tables only. */
  const float* rope_cs;
  /* LayerNorm of the finished output row in the epilogue (the "norm" that follows an attention's output projection + identity in
   * detrex's BaseTransformerLayer): C = LN(acc + bias + residual) * ln_w + ln_b over the N channels, statistics on the fp32 sums.
   * Implemented by the K = N = 256 register-resident kernel (16-bit operands, M >= 2048, plain epilogue); anything else is an
   * argument error -- callers launch ape_hip_layernorm instead.  ln_w NULL = off. */
  const float* ln_w;
  const float* ln_b;
  float ln_eps;
  int32_t reserved0;      /* callers pass 0 (the library uses the field of ITS copy for a grid hint of the K = 256 kernel) */
  /* implicit-GEMM 3 x 3 convolution (stride 1, zero padding 1; SimpleFeaturePyramid's / the mask head's 3 x 3 convs at 256 channels,
   * vit_eva_clip.py:806-842, deformable_detr_segm_vl.py:728-750): conv_h > 0 makes A the conv INPUT -- a token-major [conv_h * conv_w,
   * lda] map of C = K / 9 = 256 channels whose row of raster pixel r is conv_perm[r] (NULL: r) -- and W the [N, 9 C] weight in
   * (ky, kx, ci) order; M = conv_h * conv_w.  The kernel stages each K tile's A operand from the shifted rows (a zero row outside the
   * image: conv_zero, >= 128 zero bytes, 16-byte aligned), so the [M, 9 C] im2col matrix is never materialised; results are
   * bit-identical to ape_hip_im2col3x3 + ape_hip_gemm.  16-bit operands, >= 200 tiles of 256 x 256 (else an argument error: callers
   * fall back to im2col). */
  const int32_t* conv_perm;
  const void* conv_zero;
  int32_t conv_h, conv_w;
  /* Round 6: the row terms of the folded LayerNorm computed BY this launch (no ape_hip_row_stats launch, no second pass over A).
   * rowstat_cols > 0 with rowscale == rowshift == NULL and colvec given: the kernel accumulates sum / sum of squares of every A row
   * it stages (fp32, from the operand fragments the main loop reads anyway; columns >= rowstat_cols of A must be zero -- the K padding),
   * and its epilogue applies  acc * rstd_m - rstd_m mean_m colvec[n] + bias[n]  with mean / var over rowstat_cols values and
   * rstd = rsqrt(var + rowstat_eps).  The statistics are single-pass (var = E[x^2] - mean^2 in fp32): meant for activations whose
   * mean is not large against their spread (the SwiGLU / attention outputs the ViT's sub-LayerNorms see, vit_eva_clip.py:129-131,
   * 258-262).  Implemented by the 256 x 128 tile kernel only (tile64 == 4, 16-bit operands, K % 64 == 0, N % 128 == 0, plain
   * epilogue: alpha 1, no mask / clamp / activation / RoPE / transpose); anything else is an argument error -- callers then launch
   * ape_hip_row_stats and pass rowscale / rowshift.  0 = off. */
  int32_t rowstat_cols;
  float rowstat_eps;
} ApeGemmArgs;
int ape_hip_gemm(const ApeGemmArgs* args, void* stream);
/* symbol of the kernel the calling thread's last ape_hip_gemm launched (measurement aid: bench.py's roofline) */
const char* ape_hip_gemm_last_kernel(void);

/* Launch metering (measurement aid: bench.py's `roofline` object and per-kernel table) -- csrc/meter.cpp.  Between ape_hip_meter_begin()
 * and ape_hip_meter_end() every kernel the CALLING THREAD launches through this library goes out with its own (start, stop) HIP event
 * pair (hipExtLaunchKernelGGL): hipEventElapsedTime of the pair is the dispatch's begin-to-end duration, the number rocprofv3's kernel
 * trace reports, free of the command-processor gaps an event pair recorded AROUND a launch includes.  Eager launches only (not inside a
 * stream capture).  ape_hip_meter_count(): launches recorded so far; ape_hip_meter_read(i, &name, &ms): kernel expression (a static
 * string, the template instantiation as written at the launch site) and duration in milliseconds of launch i (waits for it). */
int ape_hip_meter_begin(void);
int ape_hip_meter_count(void);
int ape_hip_meter_end(void);
int ape_hip_meter_read(int i, const char** name, float* ms);
/* zero-fill of a device buffer on a stream (hipMemsetAsync; captured as a memset node): the padded V^T operand buffers of the forward --
 * replaces the tensor library's fill kernel on the path (torch.zeros in rounds 1-4) -- csrc/meter.cpp */
int ape_hip_zero(void* ptr, size_t nbytes, void* stream);
/* Result transfer on a DMA ENGINE (round 6; csrc/hostcopy.cpp).  `DefaultPredictor` / `inference_on_dataset` hand `Instances` to the host
 * (ape/modeling/ape_deta/deformable_detr_segm_vl.py:599-613: `.to("cpu")` of [k, H, W] masks = 105 MB per 1024^2 image).  hipMemcpyAsync runs
 * that copy as a shader blit which holds CUs for its whole PCIe-bound duration; this entry hands it to the HSA runtime's copy engines
 * instead (hsa_amd_memory_async_copy between the agents owning the two allocations).  BLOCKING and NOT stream-ordered: call it from a
 * helper thread once the source is complete.  host_dst: pinned host memory (hipHostMalloc); dev_src: device memory.
 * ape_hip_sdma_usable: 1 when the pointer pair qualifies. */
int ape_hip_sdma_usable(const void* host_dst, const void* dev_src);
int ape_hip_sdma_d2h(void* host_dst, const void* dev_src, size_t nbytes);
/* n such copies in flight at once, each split into `parts` pieces (parts <= 0: APE_SDMA_PARTS or 2).  With APE_SDMA_ENGINES=1 in the environment
 * (ape_amd/runtime.py sets it) the pieces are placed on DIFFERENT copy engines (hsa_amd_memory_async_copy_on_engine over the engines
 * hsa_amd_memory_copy_engine_status reports free for the direction); otherwise they are plain concurrent copies, which the runtime queues on one
 * engine.  Inside a running pipeline one engine moves ~37 GB/s: the 2 x 1.18 GB of masks per step of the 1536^2 / top-500 configuration need two
 * (profiles/r06_config5_transfer.txt).  Blocking like ape_hip_sdma_d2h.  ape_hip_sdma_engines: engines free now (-1 unknown). */
int ape_hip_sdma_d2h_multi(int n, void* const* host_dst, const void* const* dev_src, const size_t* nbytes, int parts);
int ape_hip_sdma_engines(const void* host_dst, const void* dev_src);

/* per-row LayerNorm statistics of x [M, C] (row stride ldx): rowscale[m] = rsqrt(var_m + eps), rowshift[m] = -mean_m * rowscale[m]
 * (biased variance, two passes) -- the row terms of the folded LayerNorm above.  -- csrc/norm.hip */
int ape_hip_row_stats(const void* x, int ldx, int dt, int M, int C, float eps, float* rowscale, float* rowshift, void* stream);

/* out[m][n] = alpha * x[m,:] . W[n,:] + bias[n], fp32 x/out, W f32 or bf16; for M <= a few rows
 * (the L=1 language side of ape/layers/fuse_helper.py:70-73,160-161). */
int ape_hip_gemv(const float* x, int ldx, const void* W, int ldw, int w_dt, const float* bias, float* out, int ldo,
                 int M, int N, int K, float alpha, void* stream);
/* the same with the element-wise tails of that language side in the epilogue (layers/fuse_helper.py; reference
 * fuse_helper.py:224-231: v + gamma_v * delta_v, l + gamma_l * delta_l): out[m][n] = scale[n] * (alpha * x[m,:] . W[n,:] + bias[n])
 * (scale may be NULL) and, when add / out2 are given, out2[m][n] = add[m][n] + out[m][n]. */
int ape_hip_gemv_affine(const float* x, int ldx, const void* W, int ldw, int w_dt, const float* bias, float* out, int ldo,
                        int M, int N, int K, float alpha, const float* scale, const float* add, int ldadd, float* out2, int ldo2,
                        void* stream);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm over the last dim: y = (x-mean)/sqrt(var+eps)*w + b  [act]  ; optional y2 = y + add
 * Replaces nn.LayerNorm / detectron2 channel-LN / ffn_ln / inner_attn_ln:
 *   vit_eva_clip.py:29-35,509,264,129 ; detectron2 get_norm("LN") used at vit_eva_clip.py:829-842;
 *   fuse_helper.py:224-225 ; detrex BaseTransformerLayer norms.
 * Columns C..Cpad-1 of y are written as zero (K padding for the next GEMM).
 * ------------------------------------------------------------------------------------------- */
typedef struct ApeLayerNormArgs {
  const void* x; /* [M, ldx] x_dt */
  const float* w; /* [C] */
  const float* b; /* [C] */
  void* y;       /* [M, ldy] y_dt */
  const void* add; /* optional [M, ldadd] add_dt: y2 = y + add */
  void* y2;        /* optional [M, ldy2] y_dt */
  int32_t M, C, Cpad;
  int32_t ldx, ldy, ldadd, ldy2;
  int32_t x_dt, y_dt, add_dt;
  int32_t act;
  float eps;
} ApeLayerNormArgs;
int ape_hip_layernorm(const ApeLayerNormArgs* args, void* stream);

/* Post-norm residual step (ape/modeling/backbone/vit_eva_clip.py:505-523 with postnorm=True, the ViT-e blocks):
 * stream[m,:] += LayerNorm(t[m,:]) * w + b on the fp32 residual stream [M, C] IN PLACE, and (copy != NULL) the new stream in
 * copy_dt (f32 | bf16) for the next linear.  t == NULL: only the copy.  C % 4 == 0, C <= 2048.  -- csrc/norm.hip */
int ape_hip_postnorm_residual(const void* t, int ldt, int t_dt, const float* w, const float* b, float eps, float* stream, int lds,
                              void* copy, int ldc, int copy_dt, int M, int C, void* hip_stream);


/* ---------------------------------------------------------------------------------------------
 * GroupNorm(G) on a token-major map x[HW, C] (NHWC): y = act(GN(x)*w + b + add)
 * Replaces nn.GroupNorm(32, 256) in detrex ChannelMapper (neck; config
 * ape_deta_vitl_eva02_clip_vlf_lsj1024_cp_16x4_1080k.py:42-55) and detectron2 get_norm("GN") in the
 * mask head (ape/modeling/ape_deta/deformable_detr_segm_vl.py:115-135, used at :741-747).
 * workspace: ape_hip_groupnorm_workspace_floats(HW, G) fp32 scratch.  C <= 256.
 * ------------------------------------------------------------------------------------------- */
typedef struct ApeGroupNormArgs {
  const void* x;   /* [HW, ldx] x_dt */
  const float* w;  /* [C] */
  const float* b;  /* [C] */
  void* y;         /* [HW, ldy] y_dt */
  const void* add; /* optional [HW, ldadd] add_dt, added after the affine, before the activation */
  float* workspace;
  int32_t HW, C, G;
  int32_t ldx, ldy, ldadd;
  int32_t x_dt, y_dt, add_dt;
  int32_t act; /* APE_ACT_NONE or APE_ACT_RELU */
  float eps;
} ApeGroupNormArgs;
int ape_hip_groupnorm_workspace_floats(int HW, int G);
int ape_hip_groupnorm(const ApeGroupNormArgs* args, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-scale deformable attention, forward only.
 * ape_hip_ms_deform_attn_forward is the replacement for the reference operator
 *   torch.ops.ape.ms_deform_attn_forward  (ape/layers/csrc/vision.cpp:76-79,
 *   ape/layers/csrc/MsDeformAttn/ms_deform_attn.h:22-40, ms_deform_attn_cuda.cu:21-81,
 *   kernel ms_deform_im2col_cuda.cuh:237-299) and of the pure-PyTorch path
 *   multi_scale_deformable_attn_pytorch (ape/layers/multi_scale_deform_attn.py:84-124).
 *   value [B, S, M, D] (row stride ldv elements per spatial position), spatial_shapes [L,2] int64
 *   (h,w), level_start_index [L] int64 -- HOST pointers here (tiny, constant per resolution; baked into the
 *   kernel argument block), device tensors in the reference operator, sampling_loc [B,Q,M,L,P,2] (x,y in [0,1]),
 *   attn_weight [B,Q,M,L,P]  ->  out [B,Q,M*D].   M = 8 heads, D = 32, P = 4, L <= 8.
 *   value/sampling_loc/attn_weight/out share `dt`.
 * ape_hip_msda_fused additionally folds multi_scale_deform_attn.py:278-311 (softmax over L*P,
 *   sampling-location arithmetic for 2-d and 4-d reference points) into the sampler:
 *   offw [B*Q, ldoffw] fp32: columns [0, M*L*P*2) raw sampling offsets, then M*L*P raw logits;
 *   ref [B*Q, L, refdim] fp32.  v_dt: 0 = f32, 1 = bf16, 2 = IEEE half values (out_dt 1 = bf16 or 0 = f32): the sampler is VALU
 *   bound and a half value needs no unpacking (one v_fma_mix_f32 per channel and corner), so the production path projects the
 *   values to half (ape_hip_gemm, out_dt = APE_DT_F16) -- 11 significant bits instead of bf16's 8.
 * ------------------------------------------------------------------------------------------- */
int ape_hip_ms_deform_attn_forward(const void* value, int ldv, const int64_t* spatial_shapes,
                                   const int64_t* level_start_index, const void* sampling_loc,
                                   const void* attn_weight, void* out, int ldout, int B, int S, int Q, int L,
                                   int dt, void* stream);
int ape_hip_msda_fused(const void* value, int ldv, int v_dt, const int64_t* spatial_shapes,
                       const int64_t* level_start_index, const float* offw, int ldoffw, const float* ref,
                       int refdim, void* out, int ldout, int out_dt, int B, int S, int Q, int L, void* stream);
/* the same with the offsets | logits stored as IEEE half (offw_f16 [B*Q, ldoffw] halves): the K = 256 GEMM that produces them
 * (ape_hip_gemm with out_dt = APE_DT_F16) is bound by the bytes it writes, and this kernel by the bytes it reads; bf16 values */
int ape_hip_msda_fused_h(const void* value, int ldv, int v_dt, const int64_t* spatial_shapes,
                       const int64_t* level_start_index, const void* offw_f16, int ldoffw, const float* ref,
                       int refdim, void* out, int ldout, int out_dt, int B, int S, int Q, int L, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Scaled-dot-product attention (non causal, no mask): O = softmax(scale * Q K^T) V
 * Replaces F.scaled_dot_product_attention at vit_eva_clip.py:261-263 (16 heads x 64, windows of
 * 1024 tokens or 4096 global) and nn.MultiheadAttention's core in the decoder self-attention
 * (detrex MultiheadAttention, deformable_transformer_vl.py:141-146; 8 heads x 32, 900 queries).
 *   Q,K: [B*N, ld] with head h at columns h*HD..; Vt: [H*HD, ldvt] = V transposed (token index
 *   b*N + key along the contiguous axis).  The bf16 kernels read Vt in 64-column tiles from each batch item's first column:
 *   every row of Vt must be readable and FINITE for columns 0 .. (B-1)*bstride + round_up(N, 64) - 1 (bstride = N here);
 *   O: [B*N, ldo].  HD in {32, 64}.
 * ------------------------------------------------------------------------------------------- */
int ape_hip_attention(const void* Q, int ldq, const void* K, int ldk, const void* Vt, int ldvt, void* O, int ldo,
                      int B, int N, int H, int HD, float scale, int dt, void* stream);
/* same, with `bstride` >= N rows between consecutive batch items (windows): batch item b owns rows b*bstride .. b*bstride+N-1
 * of Q / K / O and columns b*bstride .. of Vt (column bound above: with a stride that is not a multiple of 64 the last batch
 * item's last tile ends up to 63 columns past B*bstride -- size Vt accordingly).  For the zero-padded 14 x 14 windows of ape/modeling/backbone/vit_eva02.py:437-458
 * (196 tokens per window, stored at a stride of 200 so that every window starts 16-byte aligned in Vt). */
int ape_hip_attention_strided(const void* Q, int ldq, const void* K, int ldk, const void* Vt, int ldvt, void* O, int ldo,
                              int B, int N, int bstride, int H, int HD, float scale, int dt, void* stream);
/* the same with a causal mask (keys after the query are masked): the CLIP text tower's attention
 * (ape/modeling/text/eva02_clip/transformer.py:474-478 with the mask built at :714-720).  bf16: HD = 64. */
int ape_hip_attention_causal(const void* Q, int ldq, const void* K, int ldk, const void* Vt, int ldvt, void* O, int ldo,
                              int B, int N, int bstride, int H, int HD, float scale, int dt, void* stream);


/* ---------------------------------------------------------------------------------------------
 * Spatial gathers that keep the whole hot path token-major (NHWC) -- csrc/spatial.hip.
 * patchify   : (image - mean)/std, zero pad, 16x16 patch rows in token order tok2raster (NULL = raster):
 *              DeformableDETRSegmVL.preprocess_image (deformable_detr_segm_vl.py:846-855) + the im2col of
 *              PatchEmbed (utils_eva02.py:208-216).  img [3,h,w] fp32, out [Ht*Wt, 768].
 * im2col3x3  : operand of a 3x3/pad-1 conv, column (ky*3+kx)*C + c; `perm` maps raster index -> source row
 *              (SimpleFeaturePyramid vit_eva_clip.py:835-842; output_conv deformable_detr_segm_vl.py:122-131).
 * maxpool2x2 : nn.MaxPool2d(2,2) (vit_eva_clip.py:822-823);  gather_rows: out[r] = x[idx[r]]
 *              (LastLevelMaxPool stride-2 subsample vit_eva_clip.py:907-912; proposal gathers
 *              deformable_transformer_vl.py:641-644).
 * ------------------------------------------------------------------------------------------- */
int ape_hip_patchify(const float* img, int h, int w, const int32_t* tok2raster, int Ht, int Wt, const float* mean3,
                     const float* std3, void* out, int ldo, int out_dt, void* stream);
int ape_hip_im2col3x3(const void* x, int ldx, const int32_t* perm, int H, int W, int C, void* out, int ldo, int dt, void* stream);
int ape_hip_maxpool2x2(const void* x, int ldx, const int32_t* perm, int H, int W, int C, void* out, int ldo, int dt, void* stream);
int ape_hip_gather_rows(const void* x, int ldx, const int32_t* idx, int n, int C, void* out, int ldo, int dt, void* stream);
int ape_hip_gather_rows_i64(const void* x, int ldx, const int64_t* idx, int n, int C, void* out, int ldo, int dt, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Greedy NMS (torchvision.ops.nms / batched_nms semantics) -- csrc/select.hip.
 * Replaces torchvision batched_nms at deformable_transformer_vl.py:592-597 and detectron2 batched_nms at
 * ape/modeling/ape_deta/fast_rcnn.py:192.
 *   nms_mask          : bit matrix mask[n][ceil(n/64)] of "i suppresses j" (IoU > thr, same group)
 *   nms_scan_segments : independent scans over contiguous segments [seg[g], seg[g+1]) of candidates that
 *                       are already sorted by descending score inside each segment
 *   nms_scan_classes  : class-agnostic boxes, per-class visiting order[c][0..n): one scan per class
 * `valid` (optional uint8) marks candidates that take part at all (score threshold).
 * ------------------------------------------------------------------------------------------- */
int ape_hip_nms_mask_words(int n);
int ape_hip_nms_mask(const float* boxes_xyxy, const int32_t* groups, int n, float iou_thr, uint64_t* mask, void* stream);
int ape_hip_nms_scan_segments(const uint64_t* mask, int n, const int32_t* seg_offsets, int num_segments, int max_segment,
                              const uint8_t* valid, uint8_t* keep, void* stream);
int ape_hip_nms_scan_classes(const uint64_t* mask, int n, const int32_t* order, int num_classes, const uint8_t* valid,
                             uint8_t* keep, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Language side of BiMultiHeadAttention for one text token (ape/layers/fuse_helper.py:89-116,140):
 * out[h,:] = sum_t softmax_t(S[t,h]) * x[t,:]   with the reference's global-max / clamp sequence.
 * S [T, 8] fp32, x [T, C]; workspace ape_hip_vl_pool_workspace_floats(T, C) fp32; sub [C] or NULL is subtracted from every
 * pooled row (pooling x - sub: the softmax weights sum to one).  -- csrc/vlpool.hip
 * ------------------------------------------------------------------------------------------- */
int ape_hip_vl_pool_workspace_floats(int T, int C);
int ape_hip_vl_pool(const float* S, int lds, const void* x, int ldx, int x_dt, int T, int C, float* workspace, const float* sub,
                    float* out, void* stream);
/* per-head matrix-vector products of the same language side (fuse_helper.py:70-73 v_proj / values_v_proj applied to one
 * pooled vector per head): out[h][n] = alpha * sum_d x[h][d] * W[h][n][d] + bias[h][n]; x [H, ldx], W [H, N, D], bias [H, N] or
 * NULL, out [H, ldo], fp32; out_bf16 [H, ldob] (may be NULL) receives a 16-bit copy in copy_dt (APE_DT_BF16 | APE_DT_F16).
 * -- csrc/vlpool.hip */
int ape_hip_head_gemv(const float* x, int ldx, const float* W, const float* bias, float* out, int ldo, int H, int N, int D,
                      float alpha, void* out_bf16, int ldob, int copy_dt, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Softmaxes of the DENSE bi-directional attention (L > 1 text tokens: phrase / expression prompts;
 * ape/layers/fuse_helper.py:84-131).  S [T, nseg*L] fp32 is the score matrix (column (head, l)); gmax points at
 * the device scalar max(S) (fuse_helper.py:89-90).  -- csrc/softmax.hip
 *   segment_softmax : out[t, (h,l)] = softmax_l( clamp(S - gmax) )                     (vision side, :131)
 *   colstats        : colmax[c], colsum[c] of clamp(S[:,c] - gmax) over the T rows         (language side, :101-116)
 *   transpose       : out[c, t] = S[t, c], or with colmax/colsum the language-side softmax over t, transposed
 * ------------------------------------------------------------------------------------------- */
int ape_hip_segment_softmax(const float* S, int lds, int T, int nseg, int L, const float* gmax, void* out, int ldo, int out_dt,
                            void* stream);
int ape_hip_colstats_workspace_floats(int T, int C);
int ape_hip_colstats(const float* S, int lds, int T, int C, const float* gmax, float* workspace, float* colmax, float* colsum,
                     void* stream);
int ape_hip_transpose(const void* S, int lds, int in_dt, int T, int C, const float* gmax, const float* colmax,
                      const float* colsum, void* out, int ldo, int out_dt, void* stream);
/* Per-query class scores of the semantic / panoptic branches (round 6; csrc/softmax.hip) -- replace the tensor-library glue of
 * ape/modeling/ape_deta/deformable_detr_segm_vl.py:1251-1271 (get_stuff_score), :891-894 (semantic class weights), :944-949 (panoptic scores)
 * and the final argmax of the label map.
 *   stuff_collapse   : [Q, K] -> [Q, K - nt + 1]: column 0 = min over the nt thing columns, then the stuff columns
 *   sem_class_weights: A[c, r] = softmax_c(sigmoid(logits[qidx[r], c]) / temp) * valid[r] for r < k, 0 for k <= r < kp (lda >= kp)
 *   pan_class_scores : score / label = max_c sigmoid (transform: of softmax_c(sigmoid / temp)); keep = valid & (max sigmoid > thresh)
 *   (valid_score: fp32 [k] detection scores, a row is valid iff its score >= 0 -- the fixed-shape detection lists mark empty slots with -1; NULL = all)
 *   argmax_labels    : int16 argmax over the class axis of [C, n] scores (class stride ld_class); class0 != NaN replaces class 0's scores */
int ape_hip_stuff_collapse(const float* logits, int ldl, int Q, int K, int nt, float* out, int ldo, void* stream);
int ape_hip_sem_class_weights(const float* logits, int ldl, const int64_t* qidx, const float* valid_score, int k, int kp, int K, float temp,
                              void* A, int lda, int out_dt, void* stream);
int ape_hip_pan_class_scores(const float* logits, int ldl, const int64_t* qidx, const float* valid_score, int k, int K, float thresh, int transform,
                             float temp, float* score, int64_t* label, int32_t* label32 /* may be NULL */, uint8_t* keep, void* stream);
int ape_hip_argmax_labels(const float* x, size_t ld_class, int C, size_t n, float class0, int16_t* out, void* stream);


/* ---------------------------------------------------------------------------------------------
 * Instance-mask post-processing of the kept detections -- csrc/masks.hip
 *   mask_upsample_bits: bilinear (align_corners=False) h0 x w0 -> S x S of n mask-logit rows, then > 0
 *                       (F.interpolate + sigmoid > 0.5, deformable_detr_segm_vl.py:569-572,605)
 *   roi_align_bits    : BitMasks.crop_and_resize(boxes, P) (roi_align aligned=True, adaptive sampling, >= 0.5)
 *                       (deformable_detr_segm_vl.py:606-608)
 *   paste_bits        : detectron2 paste_masks_in_image at the output resolution, threshold 0.5
 *                       (detector_postprocess, deformable_detr_segm_vl.py:869-871)
 * ------------------------------------------------------------------------------------------- */
int ape_hip_mask_upsample_bits(const void* logits, int ldl, int dt, int h0, int w0, int S, int n, uint8_t* out, void* stream);
int ape_hip_roi_align_bits(const uint8_t* bits, int H, int W, const float* boxes, int n, int P, uint8_t* out, void* stream);
int ape_hip_paste_bits(const uint8_t* masks, int P, const float* boxes, int n, int Ho, int Wo, uint8_t* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Semantic branch (deformable_detr_segm_vl.py:628-666, _postprocess_semantic :875-918) -- csrc/masks.hip
 *   mask_upsample_sigmoid: out[(y,x), q] = sigmoid(bilinear_up(logits)[q, y, x]) for y < crop_h, x < crop_w of the
 *                          S x S upsampled grid; logits are PIXEL-MAJOR [h0*w0, n].  einsum("qc,qhw->chw") is then one
 *                          ape_hip_gemm of [K, n] x [crop_h*crop_w, n]^T.
 *   bilinear_resize      : sem_seg_postprocess's F.interpolate(bilinear, align_corners=False) [C,h,w] -> [C,H,W] fp32
 * ------------------------------------------------------------------------------------------- */
int ape_hip_mask_upsample_sigmoid(const void* logits, int ldl, int in_dt, int h0, int w0, int S, int crop_h, int crop_w, int n,
                                  void* out, int ldo, int out_dt, void* stream);
int ape_hip_bilinear_resize(const float* in, int ld_channel, int ld_row, int h, int w, int C, float* out, int H, int W,
                            void* stream);

/* ---------------------------------------------------------------------------------------------
 * Panoptic merge on the device (_postprocess_panoptic, deformable_detr_segm_vl.py:921-998) -- csrc/masks.hip.  No host round trip
 * (the reference reads three `.item()`s per kept query), so the merge can sit inside a captured step.
 *   panoptic_pixels: masks [k, h, w] fp32 mask LOGITS at the input resolution (query / row strides given), scores [k], keep [k] ->
 *                    per pixel of the H x W output frame: owner = first argmax over the kept queries of score_q * sigmoid(
 *                    bilinear(masks_q)) (int16, -1 = no kept query), conf = that probability >= prob; areas [k, 3] = (pixels owned,
 *                    pixels with p_q >= prob, both) -- zeroed by the call
 *   panoptic_decide: the sequential walk over the queries (:963-995): areas, classes [k], keep [k], isthing [num_classes] ->
 *                    seg_id [k] (0 = dropped), info [k, 3] = (id, isthing, category_id) for the first *count segments; stuff_offset
 *                    >= 0: category_id of a stuff segment = class - stuff_offset + 1 (the "things"-first stuff vocabulary, :985-986)
 *   panoptic_write : panoptic_seg [H, W] int32 = seg_id[owner] where conf, else 0
 * ------------------------------------------------------------------------------------------- */
int ape_hip_panoptic_pixels(const float* masks, int ld_query, int ld_row, int h, int w, int k, const float* scores, const uint8_t* keep,
                            float prob, int H, int W, int16_t* owner, uint8_t* conf, int* areas, void* stream);
int ape_hip_panoptic_decide(const int* areas, const int* classes, const uint8_t* keep, int k, const uint8_t* isthing, int num_classes,
                            double overlap_threshold, int stuff_offset, int* seg_id, int* info, int* count, void* stream);
int ape_hip_panoptic_write(const int16_t* owner, const uint8_t* conf, const int* seg_id, int H, int W, int* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Decoder box refinement (deformable_transformer_vl.py:203-210, 232-246):
 * new_ref = sigmoid(delta + inverse_sigmoid(ref, eps)), ref_in[q, l, :] = new_ref[q, :] * vr4[l, :].
 * delta may be NULL (new_ref = ref).  -- csrc/boxes.hip
 * ------------------------------------------------------------------------------------------- */
int ape_hip_box_refine(const float* delta, int ldd, const float* ref, const float* vr4, int L, int Q, float eps, float* new_ref,
                       float* ref_in, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Query initialisation of the two-stage decoder (deformable_transformer_vl.py:412-420 get_proposal_pos_embed, :629-645).
 * query_init: coords [T,4] fp32 unactivated boxes, topk [Q] int64 selected tokens -> reference [Q,4] = sigmoid(coords[topk]),
 *   pe [Q, 4P] (f32 | bf16) = the sine embedding (dim_t [P], scale = 2 pi), topk32 [Q] int32 copy of the indices (may be NULL).
 * query_finish: pos [Q, 2E] = pos_trans(pe), pix [Q, E] = pix_trans(output_memory[topk]) (fp32 GEMM outputs) ->
 *   query_pos = LN_pos(pos)[:, :E], query = LN_pos(pos)[:, E:] + LN_pix(pix), query_sum = query + query_pos, all [Q, E] in
 *   out_dt (f32 | bf16), E a multiple of 64, <= 512.  -- csrc/boxes.hip
 * ------------------------------------------------------------------------------------------- */
/* Detection records of one image (detector_postprocess deformable_detr_segm_vl.py:857-872; detectron2 Boxes.scale / clip /
 * nonempty): boxes * frame[0:4] clipped to [0, frame[4:8]], keep = score >= 0 and non-empty; rec [k,8] = (box, score | -1, class,
 * query, keep) with the KEPT rows first (stable partition), boxes_out [k,4] and order [k] (source row of each output row) alike.
 * frame [8] fp32 on the device = (sx, sy, sx, sy, width, height, width, height).  -- csrc/boxes.hip */
int ape_hip_det_records(const float* boxes, const float* scores, const int64_t* classes, const int64_t* query, const float* frame,
                        int k, float* rec, float* boxes_out, int32_t* order, void* stream);
int ape_hip_query_init(const float* coords, const int64_t* topk, int T, const float* dim_t, int P, float scale, int Q,
                       float* reference, void* pe, int ldpe, int pe_dt, int32_t* topk32, void* stream);
int ape_hip_query_finish(const float* pos, int ldpos, const float* pix, int ldpix, int Q, int E, const float* wpos,
                         const float* bpos, float eps_pos, const float* wpix, const float* bpix, float eps_pix, void* query_pos,
                         void* query, void* query_sum, int ldo, int out_dt, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-image-size constants of the deformable encoder for an (h, w) image inside the S x S pad, written straight into the
 * fixed buffers a captured graph reads: padding masks (deformable_detr_segm_vl.py:382-388), sine position embedding
 * (detrex PositionEmbeddingSine, ape_deta_r50.py:35-40) + level embedding (deformable_transformer_vl.py:461) as
 * lvl_pos [T, 2*npf] (f32 / bf16), valid ratios (:402-410), encoder reference points [T, L, 2] (:371-400), logit-space
 * anchors [T, 4] with +inf where unusable and the unusable mask (:321-369), box limits (w, h, w, h).
 * level_hw: HOST int [L, 2]; dim_t: device [npf] = temperature ** (2 * (i / 2) / npf).  -- csrc/geometry.hip
 * ------------------------------------------------------------------------------------------- */
int ape_hip_geometry(int S, int h, int w, int L, const int* level_hw, const float* dim_t, int npf, const float* level_embeds,
                     float offset, float eps, float scale, void* lvl_pos, int lvl_pos_dt, uint8_t* mask_u8, uint8_t* mask_bool,
                     uint8_t* invalid_u8, float* enc_ref, float* proposals, float* valid_ratios, float* vr4, float* box_scale,
                     void* stream);

/* ---------------------------------------------------------------------------------------------
 * Input pipeline: the predictor's test-time resize, ape/engine/defaults.py:213-222
 * (`self.aug.get_transform(img).apply_image(img)` = ResizeShortestEdge -> PIL Image.resize(BILINEAR) on uint8), bit
 * exact with Pillow's Resample.c (separable triangle filter, 22-bit fixed-point coefficients, horizontal pass to uint8,
 * vertical pass to uint8).  -- csrc/imageio.hip
 *   resize_coeffs    : HOST function.  Coefficients of one axis (Pillow precompute_coeffs + normalize_coeffs_8bpc):
 *                      bounds [out, 2] = (first source index, count), kk [out, ksize_cap] int32.  Returns ksize (also with
 *                      bounds = kk = NULL, to size the arrays); the caller uploads both arrays to the device.
 *   resize_tile_rows : HOST function.  From the HOST copy of the vertical bounds: output-tile height (*tile_h) and the
 *                      LDS rows (return value) the kernel needs.
 *   resize_bilinear_u8: src HWC uint8 [H, src_ld bytes] -> dst_kind 0: uint8 HWC (dst_ld bytes per row), 1: float32 CHW
 *                      (dst_ld floats per row, dst_plane floats per channel) = the model's `image` input
 *                      (defaults.py:220 `image.astype("float32").transpose(2, 0, 1)`); flip = 1 reverses the channel order
 *                      (defaults.py:216 BGR -> RGB).  bounds_* / kk_* are DEVICE arrays; NULL = that axis keeps its size.
 * ------------------------------------------------------------------------------------------- */
int ape_hip_resize_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk, int ksize_cap);
int ape_hip_resize_tile_rows(const int32_t* bounds_v_host, int newh, int* tile_h);
int ape_hip_resize_bilinear_u8(const uint8_t* src, int H, int W, int src_ld, const int32_t* bounds_h, const int32_t* kk_h,
                               int ks_h, const int32_t* bounds_v, const int32_t* kk_v, int ks_v, int newh, int neww, int tile_h,
                               int lds_rows, void* dst, int dst_kind, int dst_ld, int dst_plane, int flip, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Evaluator wire format: COCO run-length encoding of the pasted instance masks, what
 * instances_to_coco_json (ape/evaluation/d3_evaluation.py:441-493, refcoco_evaluation.py:425-477, demo/demo_lazy.py:189-198)
 * obtains from pycocotools `mask_util.encode(np.array(mask[:, :, None], order="F"))` (cocoapi maskApi.c rleEncode /
 * rleToString).  -- csrc/imageio.hip
 *   rle_encode   : masks [n, H, W] uint8 (0 / non-zero, row-major) -> counts [n, cap] uint32 = column-major run lengths
 *                  starting with the zeros, nruns [n].  nruns[i] > cap means the encoding was truncated (call again with a
 *                  larger cap).  workspace: ape_hip_rle_workspace_words(n, H, W, cap) uint32.
 *   rle_to_string: HOST function, counts -> the ASCII string stored under "counts"; returns its length (-needed if the
 *                  buffer is too small).
 * ------------------------------------------------------------------------------------------- */
int ape_hip_rle_workspace_words(int n, int H, int W, int cap);
int ape_hip_rle_encode(const uint8_t* masks, int n, int H, int W, uint32_t* workspace, uint32_t* counts, int cap, uint32_t* nruns,
                       void* stream);
int ape_hip_rle_to_string(const uint32_t* counts, int n, char* out, int cap);

/* ---------------------------------------------------------------------------------------------
 * Text tower, token embedding: ape/modeling/text/eva02_clip/transformer.py:724-726 /
 * clip_wrapper_eva02.py:136-138  x = token_embedding(text) + positional_embedding.
 * tokens [B, ldt] int32; table [vocab, ldtab], pos [ctx, ldpos] (dt f32 / bf16); out [B * Lp, ldo] fp32 with rows
 * b * Lp + t: the first L positions of every text, zero rows for L <= t < Lp (Lp = the attention's batch stride).
 * The rest of the tower runs on ape_hip_gemm / ape_hip_layernorm / ape_hip_attention_causal.  -- csrc/spatial.hip
 * ------------------------------------------------------------------------------------------- */
int ape_hip_embed_tokens(const int32_t* tokens, int ldt, const void* table, int ldtab, const void* pos, int ldpos, int dt, float* out,
                         int ldo, int B, int L, int Lp, int W, int vocab, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Data-dependent selections as fixed-shape device code -- csrc/topk.hip.  Radix select over order-preserving keys + stable
 * compaction (ties: lowest index) + bitonic sort in LDS by 1024-thread workgroups; large arrays in two stages (one workgroup per
 * register-resident chunk, then one workgroup per problem over the chunks' lists; `workspace`: ape_hip_topk_workspace_words).
 *
 * Encoder proposals (ape/modeling/ape_deta/deformable_transformer_vl.py:503-533, 565-627); level l owns tokens
 * level_start[l] .. +level_n[l] (HOST int arrays, L <= 5):
 *   enc_finalize   : per token the (logit, box delta) pair of the larger logit of the two head copies (first on ties), + anchors
 *                    -> enc_class [T], enc_coord [T,4] (unactivated), xyxy [T,4] = clamp(corners(sigmoid(enc_coord)), 0, 1)
 *   proposal_topk  : cand [L*k]: per level the top k of sigmoid(logit) * level mask over ALL T tokens (ties: lowest index;
 *                    a level with fewer than k tokens continues with the lowest-index tokens of the other levels, whose
 *                    masked score is 0); alt [k_alt]: top k_alt of the raw logits over all tokens (the :600-606 fallback)
 *   proposal_order : the n = L*k candidates by descending logit (stable) = "A order": cand_a, lv_a (token's level), pos_b =
 *                    position in the level-major "B order"; boxes_b / groups_b / seg [L+1] are the NMS segments in B order
 *                    (feed ape_hip_nms_mask + ape_hip_nms_scan_segments)
 *   proposal_quota : keep_b = NMS survivors in B order -> out [nq] int64: per level the first nq / L survivors in A order,
 *                    then the best of the rest; with fewer than nq survivors the first n_alt entries of alt take the
 *                    candidates' place; zero padded
 * Final detections (ape/modeling/ape_deta/fast_rcnn.py:97-201, deformable_detr_segm_vl.py:759-810):
 *   det_sort       : logits [Q, ldl] (K classes), boxes [Q,4] cxcywh, scale [4] device (w,h,w,h) -> xyxy [Q,4] scaled + clipped
 *                    (zero rows where box or scores are not finite), finite [Q]; per class the queries by descending sigmoid
 *                    score (stable): sorted [K,Q], order [K,Q] int32, valid [K,Q] = score > thresh & finite.  Q <= 1024.
 *   det_topk       : keep [K,Q] = class-wise NMS survivors in visiting order (ape_hip_nms_scan_classes) -> the k best
 *                    (score, class, query, box); suppressed pairs rank as score -1 (ties: lowest (class, rank))
 * ------------------------------------------------------------------------------------------- */
int ape_hip_enc_finalize(const float* cls2, const float* d, const float* anchors, int T, float* enc_class, float* enc_coord,
                         float* xyxy, void* stream);
int ape_hip_topk_workspace_words(int n_total); /* uint64 words of `workspace` for proposal_topk (any T) / det_topk (n_total = K*Q) */
int ape_hip_proposal_topk(const float* logit, int T, const int* level_start, const int* level_n, int L, int k, int k_alt,
                          uint64_t* workspace, int32_t* cand, int32_t* alt, void* stream);
int ape_hip_proposal_order(const int32_t* cand, int n, const float* logit, const float* xyxy, const int* level_start,
                           const int* level_n, int L, float* boxes_b, int32_t* groups_b, int32_t* seg, int32_t* cand_a,
                           int32_t* lv_a, int32_t* pos_b, void* stream);
int ape_hip_proposal_quota(const int32_t* cand_a, const int32_t* lv_a, const int32_t* pos_b, const uint8_t* keep_b, int n,
                           const int32_t* alt, int n_alt, const int* level_start, const int* level_n, int L, int nq, int64_t* out,
                           void* stream);
int ape_hip_det_sort(const float* logits, int ldl, int Q, int K, const float* boxes, const float* scale, float thresh, float* xyxy,
                     uint8_t* finite, float* sorted, int32_t* order, uint8_t* valid, void* stream);
int ape_hip_det_topk(const float* sorted, const uint8_t* keep, const int32_t* order, const float* xyxy, int K, int Q, int k,
                     uint64_t* workspace, float* det_boxes, float* det_scores, int64_t* det_classes, int64_t* det_query, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Attention with a q.k width HDQ different from the V / output width HDV (HDQ = 256 | 288 | 320, HDV = 128), and the kernel that
 * builds such operands for the EVA-01 MIM ViT's decomposed relative positions (ape/modeling/backbone/vit_eva.py:121-146,
 * utils_eva.py:132-161):  attn[q, (kh, kw)] = scale q.k + q.Rh[qh - kh] + q.Rw[qw - kw] = q_ext . k_ext  with
 *   q_ext = [scale q (hd) | q.Rh[qh - kh], kh = 0 .. Hk-1 | q.Rw[qw - kw], kw = 0 .. Wk-1 | 0 ...],  k_ext = [k | one-hot(kh) | one-hot(kw) | 0 ...]
 * ape_hip_relpos_extend: q, k [rows, ldqk] (head h at columns h * hs), t [rows * tper, ldt] = q . [Rh ; Rw]^T (2 Hk - 1 + 2 Wk - 1
 * columns, from ape_hip_gemm; the row of (token, head) is token * tper + head, tper >= nh), ty / tx [period] = position of token (row %% period) inside its attention group ->
 * q_ext, k_ext [rows, lde] (head h at columns h * hdq), all tensors in dtype dt (0 / 1 / 2).  -- csrc/relpos.hip, csrc/attention.hip
 * ------------------------------------------------------------------------------------------- */
int ape_hip_attention_ext(const void* Q, int ldq, const void* K, int ldk, const void* Vt, int ldvt, void* O, int ldo, int B, int N,
                          int bstride, int H, int HDQ, int HDV, float scale, int dt, void* stream);
int ape_hip_relpos_extend(const void* q, const void* k, int ldqk, const void* t, int ldt, int tper, const int* ty, const int* tx, int period,
                          void* q_ext, void* k_ext, int lde, int rows, int nh, int hs, int hd, int Hk, int Wk, int hdq, float scale, int dt,
                          void* stream);

/* ---------------------------------------------------------------------------------------------
 * The encoder / decoder FFN in one kernel: y = residual + relu(x W1^T + b1) W2^T + b2 (detrex FFN with add_identity,
 * ape/modeling/ape_deta/deformable_transformer_vl.py:45-54 and :160-166), 16-bit in / out (dt = APE_DT_BF16 | APE_DT_F16: x, the
 * weights, the residual, y and the hidden activation between the two contractions all in that type), K = N = 256, HID %% 64 == 0 (<= 4096).
 * The hidden activations stay in registers as the B operand of the second MFMA (csrc/ffn_fused.hip): the [M, HID] tensor the
 * two-GEMM form writes and reads back (357 MB per encoder layer at 1024^2) never exists.  residual may be NULL.
 * w2_permuted != 0: W2's hidden columns are stored pre-permuted inside every group of 32 -- position 8 g + e holds hidden
 * 4 g + e (e < 4) / 16 + 4 g + e - 4 (e >= 4), g = 0..3 -- the k order of the second MFMA (ape_amd.packing.permute_ffn_w2).
 * ln_weight / ln_bias (fp32 [256], both or neither): y = LayerNorm(residual + ffn(x)) over the 256 channels with eps = ln_eps -- the
 * "norm" that follows the "ffn" in detrex's BaseTransformerLayer -- on the fp32 sums, in the same launch.
 * ------------------------------------------------------------------------------------------- */
int ape_hip_ffn_fused(const void* X, int ldx, const void* W1, int ldw1, const float* b1, const void* W2, int ldw2, const float* b2,
                      const void* residual, int ldr, void* Y, int ldy, int M, int K, int HID, int N, int w2_permuted, int dt,
                      const float* ln_weight, const float* ln_bias, float ln_eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* APE_HIP_H */
