/*
 * vd_hip.h - C ABI of libvd_hip.so, the MI355X (gfx950 / CDNA4) kernel library behind the
 * Versatile-Diffusion sampling hot path.
 *
 * The reference (SHI-Labs/Versatile-Diffusion) has no native layer: every FLOP of the path is a
 * stock torch op issued from Python.  This header is therefore the boundary the *new* Python
 * modules (versatile-diffusion_amd/lib/model_zoo/...) bind with ctypes; each entry point names
 * the reference code whose arithmetic it replaces (paths relative to /root/reference).
 *
 * Conventions
 *   - all pointers are device pointers owned by the caller (torch caching allocator); nothing is
 *     allocated, freed or synchronised inside the library; every launch goes to `stream`
 *   - activations are fp16, channels-last: [B, H, W, C] == (B*H*W, C) row-major
 *   - weights are fp16, K-contiguous: Linear [out][in] as torch stores it, conv repacked once at
 *     load time to [Cout][kh][kw][Cin]
 *   - statistics (GroupNorm, LayerNorm, softmax) and MFMA accumulation are fp32
 *   - return value: 0 on success, negative VD_ERR_* otherwise; vd_last_error() returns the
 *     thread-local message of the last failure
 *   - threads: entry points may be called concurrently from several host threads (one stream per
 *     thread).  Process-wide state is limited to the optional tuned launch table (vd_gemm_tune_*,
 *     mutex-protected), the developer tile override (atomic), per-device kernel attributes
 *     (atomic bit mask, idempotent) and development switches read once from the environment
 *     (VD_GEMM_TILE / VD_CONV_HALO / VD_ATTN_PIPE: function-local statics, initialised under the
 *     C++11 guarantee).  Scratch (workspaces, split-K counters) is caller-owned, one set per stream.
 *
 * ABI 8 (round 6) REMOVED entry points and options that had lost every measurement (vd_conv3x3_wreg_*,
 * vd_gemm_groupnorm_ok, VD_EPI_GROUPNORM / VD_EPI_GN_SILU: rejected, VdGemmDesc.gn_*: reserved); nothing
 * was added, and no struct layout changed.
 */
#ifndef VD_HIP_H
#define VD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t; /* same declaration as <hip/hip_runtime_api.h>: plain C hosts need no HIP headers */

#define VD_HIP_ABI_VERSION 8
#define VD_MAX_SPLIT_K 32

/* ---- epilogue description for vd_gemm_f16 ------------------------------------------------ */
#define VD_EPI_BIAS 1         /* + bias[n]                                                    */
#define VD_EPI_ROWVEC 2       /* + rowvec[m / rows_per_batch][n]  (ResBlock "h + emb_out")    */
#define VD_EPI_RESIDUAL 4     /* + res[m][n] after activation and alpha                       */
#define VD_EPI_BIAS_ALONG_M 8 /* bias indexed by output row (V^T = Wv x^T in the VAE AttnBlock) */
#define VD_EPI_OUT_F32 16     /* store fp32 instead of fp16                                   */
#define VD_EPI_LNFOLD 32      /* A rows are LayerNorm'ed on the fly, see VdGemmDesc.colsum     */
#define VD_EPI_LN_INLOOP 64   /* with VD_EPI_LNFOLD and ln_stats == NULL: row statistics inside the K loop (explicit opt-in) */
#define VD_EPI_GROUPNORM 128  /* reserved (ABI 5-7: GroupNorm inside the split-K reduce; removed in ABI 8, rejected)              */
#define VD_EPI_GN_SILU 256    /* reserved, as above                                                                        */
#define VD_EPI_LN_SUMS 512    /* with VD_EPI_LNFOLD (ABI 6): ln_stats points at the int64 [M][2] fixed-point (sum, sum of squares)
                               * of every A row that a producer launch accumulated through VdGemmDesc.row_sums, instead of fp32
                               * (mean, rstd); the epilogue derives mean / rstd over K with ln_eps */

#define VD_ACT_NONE 0
#define VD_ACT_GEGLU 1      /* out[:, j] = val_j * gelu_erf(gate_j); W/bias packed per 64 rows as [32 val | 32 gate] */
#define VD_ACT_QUICK_GELU 2 /* x * sigmoid(1.702 x)  (HF CLIP)                                 */
#define VD_ACT_SILU 3
#define VD_ACT_GELU_TANH 4  /* 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))  (GPT-2 MLP of the Optimus decoder) */

/*
 * One fused GEMM / implicit-GEMM convolution:
 *    out[m][n] = ( act( sum_k A[m][k] W[n][k] + bias + rowvec ) ) * alpha + res[m][n]
 * With VD_EPI_LNFOLD the A rows are layer-normalised over K on the fly:
 *    LN(a)[m][k] = (a[m][k] - mean_m) * rstd_m * gamma[k] + beta[k]
 *    sum_k LN(a)[m][k] W[n][k] = rstd_m * ( sum_k a[m][k] W'[n][k] - mean_m * colsum[n] ) + bias'[n]
 * with W' = gamma (*) W (passed as `w`), colsum[n] = sum_k W'[n][k] (fp32) and bias' = beta W^T + bias (passed as `bias`),
 * all prepared once per layer by the host.  (mean_m, rstd_m) are read from `ln_stats` (vd_row_stats_f16: two-pass
 * statistics, one extra read of A).  With VD_EPI_LN_INLOOP and `ln_stats` == NULL the kernel instead accumulates sum and sum
 * of squares of every A row from the operand fragments inside its K loop (biased variance, `ln_eps`; one-pass E[x^2] - mean^2
 * on the raw fp16 values, less robust for |mean| >> std, hence opt-in: a NULL `ln_stats` without the flag is an error), so
 * the nn.LayerNorm in front of a projection (lib/model_zoo/attention.py:214-218) costs no pass over memory and no launch.
 * Plain (non-conv, single-source) A only, no split-K.
 * A[m][k] is gathered on the fly: m -> (b, oy, ox) over Hout x Wout, k -> (ky, kx, c) with c running
 * over the channels of a0 (c0) then a1 (c1) -- i.e. torch.cat([a0, a1], dim=1) is never materialised --
 * at input pixel ((oy*stride - pad + ky) >> ups, (ox*stride - pad + kx) >> ups) (ups=1: nearest 2x upsample
 * fused in front of the conv).  Plain matrices: set Hout = Wout = 0.
 */
typedef struct VdGemmDesc {
    const void* a0;      /* fp16 activation source 0, row stride lda0 (elements)                 */
    const void* a1;      /* optional fp16 source 1 (channel concat), row stride lda1             */
    const void* w;       /* fp16 [N][ldw]                                                        */
    const void* bias;    /* fp16 [N] (or [M] with VD_EPI_BIAS_ALONG_M)                            */
    const void* rowvec;  /* fp16 [M / rows_per_batch][N]                                          */
    const void* res;     /* fp16 [M][ldr]                                                        */
    void* out;           /* fp16 (or fp32) [M][ldc]                                              */
    float* ws;           /* split-K workspace (vd_gemm_workspace_bytes), may be NULL             */
    int32_t M, N, K;
    int32_t c0, c1, lda0, lda1, ldw, ldc, ldr;
    int32_t Hin, Win, Hout, Wout, ksize, stride, pad, ups;
    int32_t rows_per_batch;
    int32_t flags, act;
    float alpha;
    int32_t batch;       /* blockIdx.z batches with the element strides below                    */
    int32_t split_k;     /* 0 = heuristic                                                        */
    int64_t stride_a, stride_w, stride_out, stride_res;
    const float* colsum; /* VD_EPI_LNFOLD: fp32 [N] row sums of w                                */
    float ln_eps;        /* VD_EPI_LNFOLD with ln_stats == NULL: epsilon of the in-loop statistics              */
    int32_t reserved;
    int32_t* sync;       /* optional split-K arrival counters: VD_GEMM_SYNC_INTS ints, ZERO before their first use and
                          * private to the stream (launches on one stream are ordered; the kernel leaves them zero).  With
                          * them the last-arriving block of each output tile sums the tile's fp32 slabs (in split order:
                          * results stay run-to-run identical) and runs the fused epilogue itself -- no reduce launch.   */
    const float* ln_stats; /* VD_EPI_LNFOLD: fp32 [batch*M][2] = (mean, rstd) of every A row from vd_row_stats_f16 (NULL only
                            * with VD_EPI_LN_INLOOP)                                                                */
    float* out_stats;      /* optional (ABI 5): per-channel statistics of the STORED output for a consuming GroupNorm, fp32
                            * [M / R][N][2] = (mean, M2 = sum (x - mean)^2) over blocks of R rows of one image; R =
                            * vd_gemm_stat_rows(desc) (depends on the launch the planner picks; 0 = this launch cannot emit
                            * them: leave out_stats NULL and use vd_chan_stats_f16).  fp16 output, batch 1, N % 8 == 0.     */
    int32_t stat_img_rows; /* rows of one image for out_stats (partials never mix images); 0 = Hout * Wout (M for plain matrices) */
    int32_t gn_groups;     /* reserved, must be 0 / NULL / 0 (ABI 5-7: operands of VD_EPI_GROUPNORM, the consuming GroupNorm inside the */
    const void* gn_gamma;  /* kernel that sums split-K slabs; measured neutral against out_stats + the apply launch and removed in   */
    const void* gn_beta;   /* ABI 8 -- the fields keep the descriptor layout)                                                       */
    float gn_eps;
    int32_t reserved3;
    /* Skip 1x1 convolution folded into a 3x3 convolution (ABI 5): out += W_s [N][skip_c0 + skip_c1] . cat(skip_a0, skip_a1)[pixel]
     * -- ResBlock's `skip_connection(x) + h` (lib/model_zoo/openaimodel.py:238-251,272-274) as extra K of the second conv
     * instead of its own GEMM and a residual round trip.  skip_a0 / skip_a1: fp16 tensors on the OUTPUT grid (row strides
     * skip_lda0 / skip_lda1, channels multiples of 64); skip_w: fp16 [N][skip_ldw], K-contiguous.  Only the halo-resident
     * convolution takes it (vd_gemm_skip_ok); its bias belongs into `bias`. */
    const void* skip_a0;
    const void* skip_a1;
    const void* skip_w;
    int32_t skip_c0, skip_c1, skip_lda0, skip_lda1, skip_ldw, reserved4;
    /* Row statistics for a consuming LayerNorm fold (ABI 6): int64 [M][2], ZEROED by the caller; every block adds (atomically)
     * the sum and the sum of squares of the fp16 values it stores over its columns of each row, in fixed point (sum x 2^24, sum
     * of squares x 2^16: integer adds, so the result does not depend on the order the blocks arrive in), so after the launch
     * row_sums[m] = (sum_n out[m][n], sum_n out[m][n]^2) -- the statistics half of the nn.LayerNorm in front of the NEXT
     * projection (lib/model_zoo/attention.py:214-218) without vd_row_stats_f16's launch and extra read.  The consumer passes the
     * buffer as ln_stats with VD_EPI_LN_SUMS.  Only unsplit gemm_f16_kernel launches with vector-aligned fp16 output take it:
     * ask vd_gemm_row_sums_ok(desc) first. */
    void* row_sums;
    /* Per-(image, channel) sums for a consuming GroupNorm (ABI 7), next to out_stats: int64 [images][N][2], ZEROED by the caller.
     * Every launch that writes an out_stats partial (mean, M2 over R rows of one image) also adds, atomically and in fixed point,
     * R * mean x 2^32 and (M2 + R * mean^2) x 2^16 (formed in fp64 from the shifted partial) for the image the partial lies in
     * (row / stat_img_rows), so after the launch stat_sums[img][n] = (sum, sum of squares) over the image's rows of channel n:
     * ONE pair per (image, channel) whatever the producer's tile shape, which the consumer folds itself (vd_gn_apply_sums_f16:
     * no vd_gn_table_f32 launch).  Integer adds: the result does not depend on the order the blocks arrive in.  Needs out_stats
     * and stat_img_rows > 0; launchers that cannot emit out_stats ignore it (the consumer checks the producer's report). */
    void* stat_sums;
} VdGemmDesc;
#define VD_GEMM_SYNC_INTS 16384

/* Replaces nn.Conv2d / nn.Linear / torch.bmm call sites:
 *   lib/model_zoo/openaimodel.py:89-117 (Upsample), :133-159 (Downsample), :254-274 (ResBlock._forward)
 *   lib/model_zoo/attention.py:37-64 (GEGLU FF), :170-193 (q/k/v/out projections), :255-266 (proj_in/out)
 *   lib/model_zoo/autokl_modules.py:42-79,82-141,150-202 (VAE convs, AttnBlock bmm)
 *   HF CLIP linear layers reached from lib/model_zoo/clip.py:58-61,95-100 */
int vd_gemm_f16(const VdGemmDesc* desc, hipStream_t stream);
size_t vd_gemm_workspace_bytes(const VdGemmDesc* desc);
/* 1 when the launch vd_gemm_f16 would plan for `desc` can accumulate VdGemmDesc.row_sums (see there), else 0. */
int vd_gemm_row_sums_ok(const VdGemmDesc* desc);
/* Dry run of the launch planner: tile_cfg indexes the instantiation table of vd_gemm_config_name(); nsplit = split-K
 * factor.  Lets bench.py attribute measured time / algorithmic FLOPs to the kernel instantiation that actually ran. */
int vd_gemm_plan(const VdGemmDesc* desc, int* tile_cfg, int* nsplit);
/* Rows per statistics partial the launch planned for `desc` would write to desc->out_stats (the pointer itself is not
 * read): the halo-resident convolution emits one partial per 256-pixel patch (or per whole small image), gemm_f16_kernel one
 * per min(tile rows, image rows), the split-K reduce one per 64 rows.  *rows = 0: no statistics from this launch. */
int vd_gemm_stat_rows(const VdGemmDesc* desc, int* rows);
/* 1 when the launch planned for `desc` takes the folded skip 1x1 convolution (skip_a0 / skip_w set): the halo-resident 3x3
 * convolution with 256 x 160 blocks, no upsample, the skip tensors on the output grid. */
int vd_gemm_skip_ok(const VdGemmDesc* desc);
/* "gemm_f16_kernel<BM,BN,WM,WN,NT,STAGES,KB>" of tile_cfg (NULL when out of range); vd_gemm_num_configs() entries. */
const char* vd_gemm_config_name(int tile_cfg);
int vd_gemm_num_configs(void);
/* Development hook (tools/gemm_sweep.py, A/B runs): force every following vd_gemm_f16 of this process onto tile_cfg
 * (-1 = planner's choice) where the shape permits.  Process-global and unsynchronised: not for production use. */
int vd_gemm_set_override(int tile_cfg);
/* 3x3 / stride 1 / pad 1 convolutions (VdGemmDesc.ksize = 3) whose output grid tiles into 8 x 32 pixel patches run on
 * conv3x3_halo_kernel: the (rows + 2) x (cols + 2) input halo of a patch is staged in LDS once per 64-channel chunk and the
 * nine taps read shifted views of it, instead of gemm_f16_kernel's nine gathers (ResBlock / Upsample / VAE convs:
 * lib/model_zoo/openaimodel.py:89-117,254-274, autokl_modules.py:82-141).  vd_gemm_plan reports these launches as
 * tile_cfg = vd_gemm_num_configs() + variant, 0 <= variant < VD_CONV_HALO_VARIANTS (vd_gemm_config_name knows them).
 * Development hook: -1 = planner's choice (default; also the environment variable VD_CONV_HALO), 0 = never (every conv on
 * gemm_f16_kernel), k > 0 = force variant k - 1 where the geometry permits.  Process-global like vd_gemm_set_override. */
#define VD_CONV_HALO_VARIANTS 13
int vd_conv_halo_set_variant(int setting);
/* Tuned launch table: a problem (M, N, K, ksize, epilogue class: bit 0 GEGLU, bit 1 LayerNorm fold, bit 2 two-source A) is
 * launched with tile_cfg / split-K nsplit (0 / 1 = none) instead of the cost model's choice.  The host loads the table
 * (lib/gemm_tune.py) from a file tools/tune_forward.py measured inside a UNet forward -- the setting that decides, since
 * weights then stream from HBM and activations come hot from the previous kernel.  Thread-safe. */
/* The 3x3 / stride 1 / pad 1 convolutions on 8x8 images (the ResBlocks at ds = 8: lib/model_zoo/openaimodel.py:254-274 with
 * 1280 / 2560 input channels) as a WEIGHT-STREAMING kernel: M = images x 64 is tiny, the layer is its 29.5 / 59 MB weight
 * stream.  `desc` as for vd_gemm_f16 (its `w` is not read; `ws` must hold vd_gemm_workspace_bytes); `w_stream` holds the same
 * weights in MFMA-fragment order, fp16 [N / 32][(c0 + c1) / 64][9 taps][4 k-steps][64 lanes][8]: lane l of the fragment of
 * (n tile t, chunk c, tap, k-step s) holds W[32 t + (l & 31)][tap][64 c + 16 s + 8 (l >> 5) .. + 8] (vd_hip/pack.py:
 * pack_conv_weight_stream) -- every A operand of an MFMA is one coalesced 1-KiB load straight into registers, the input halo
 * of a chunk is staged in LDS once, K is split over chunks (fp32 slabs + the reduce kernel, which runs desc's epilogue and
 * emits desc->out_stats in partials of 64 rows when asked).  A folded skip convolution (desc->skip_a0 / skip_a1 / skip_c0 /
 * skip_c1) is taken too: desc->skip_w then holds the 1x1 weights in fragment order, fp16 [N / 32][(skip_c0 + skip_c1) / 64][4]
 * [64 lanes][8] (pack_linear_weight_stream).  vd_conv3x3_wstream_supported: 1 when desc's geometry fits. */
int vd_conv3x3_wstream_f16(const VdGemmDesc* desc, const void* w_stream, hipStream_t stream);
/* vd_gemm_wstream_f16: plain GEMM y = x W^T (+ fused epilogue of the descriptor) for long-K, small-M projections -- the output
 * projection of the gated feed-forward at the 16x16 / 8x8 levels (lib/model_zoo/attention.py:37-64, FeedForward.net[2]) -- with
 * the weights in MFMA-fragment order (w_stream: fp16 [N / 32][K / 64][4][64 lanes][8], pack_linear_weight_stream) streamed
 * straight into registers; only the activation tile takes the LDS path.  Split over K in 64-deep chunks + the split-K reduce
 * (needs desc->ws, vd_gemm_workspace_bytes).  vd_gemm_wstream_supported: 1 when desc's shape fits (single source, M % 128 ==
 * 0, N % 256 == 0, K % 64 == 0, plain fp16 epilogue). */
int vd_gemm_wstream_supported(const VdGemmDesc* desc);
int vd_gemm_wstream_f16(const VdGemmDesc* desc, const void* w_stream, hipStream_t stream);
int vd_conv3x3_wstream_supported(const VdGemmDesc* desc);
/* Split factors the two weight-streaming launchers will use for `desc` (ABI 6): set VdGemmDesc.split_k to *nsplit before sizing
 * the workspace with vd_gemm_workspace_bytes.  vd_conv3x3_wstream_plan answers 0 when the launch goes to the whole-K kernel
 * (conv_wsk_kernel.h, round 5: a block owns 128 pixels x 32 channels for the whole K, its four waves split the k-steps, the
 * fused epilogue and the GroupNorm statistics run in the kernel): no workspace, no reduce launch. */
int vd_conv3x3_wstream_plan(const VdGemmDesc* desc, int* nsplit);
int vd_gemm_wstream_plan(const VdGemmDesc* desc, int* nsplit);
/* Development hook: kernel instance (0 = default) and the grid size the split over chunks aims for (256). Process-global. */
int vd_conv3x3_wstream_set_variant(int variant, int target_blocks);
int vd_gemm_tune_set(int M, int N, int K, int ksize, int epi_class, int tile_cfg, int nsplit);
int vd_gemm_tune_clear(void);

/* y[M][N] = [LayerNorm](x[M][320]) W[N][320]^T + bias (+ res) with the block's 128 rows of x resident in registers (no activation
 * tile, no activation DMA): the K = 320 projections of the UNet's 64x64 level (proj_in / proj_out / to_out, N = 320; the fused
 * q | k | v projection, N = 960, with the LayerNorm applied in registers).  x: fp16 [M][320] contiguous; w: fp16 [N][320]
 * (gamma-folded when layernorm != 0), bias: fp16 [N] or NULL (beta W^T + bias when layernorm != 0), res: fp16 [M][N] or NULL,
 * y: fp16 [M][N]; N a multiple of 320 (vd_gemm_row320_supported).  Same result as vd_gemm_f16 on the same operands
 * (VD_EPI_BIAS | VD_EPI_RESIDUAL | VD_EPI_LNFOLD) up to fp16 rounding of the normalised rows.
 * Replaces nn.Linear / 1x1 nn.Conv2d of lib/model_zoo/attention.py:159-163,170-193,245-258 at inner width 320. */
int vd_gemm_row320_f16(const void* x, const void* w, const void* bias, const void* res, void* y, int64_t M, int N,
                       int layernorm, float ln_eps, hipStream_t stream);
int vd_gemm_row320_supported(int64_t M, int N, int K);

/* The entry of a SpatialTransformer at inner width 320 in one launch (after the statistics): GroupNorm applied as a per-sample
 * affine map -> proj_in -> h (the residual stream, written out), LayerNorm(h) -> fused q | k | v projection:
 *     h = (x * scale[img] + shift[img]) W1^T + b1,     y2 = LayerNorm(h) W2^T + b2
 * gn_center (ABI 7, may be NULL): fp16 [M / rows_per_image][320] = fp16(group mean) per channel; the map is then applied as
 * (x - center) * scale + shift with shift = beta - (mean - center) * scale (vd_gn_affine_from_stats_f16 with `center`): x - center is
 * exact in fp16 near the mean and both operands are O(1), so the fp16 map loses 2^-11 of the NORMALISED value instead of
 * |mean| / sigma * 2^-11.
 * x, h: fp16 [M][320]; gn_scale / gn_shift: fp16 [M / rows_per_image][320] from vd_groupnorm_affine_f16; w1: fp16 [320][320],
 * b1: fp16 [320]; w2: fp16 [N2][320] with the LayerNorm's gamma folded in, b2: fp16 [N2] = beta W2^T (or NULL); y2: fp16
 * [M][N2], N2 a multiple of 320; rows_per_image a multiple of 128.  Replaces Normalize -> proj_in (lib/model_zoo/
 * attention.py:236-258), norm1 and to_q / to_k / to_v (:214, :170-176) and this library's vd_groupnorm_silu_f16 (apply pass) ->
 * vd_gemm_f16 -> vd_gemm_row320_f16 chain. */
int vd_gemm_row320_chain_f16(const void* x, const void* gn_scale, const void* gn_shift, const void* gn_center, int rows_per_image, const void* w1,
                             const void* b1, void* h, const void* w2, const void* b2, void* y2, int64_t M, int N2,
                             float ln_eps, hipStream_t stream);
/* GroupNorm statistics as a per-(sample, channel) affine map: scale = rstd * gamma, shift = beta - mean * scale (fp16 [B][C]),
 * for consumers that apply the normalisation themselves.  stats: fp32 scratch of vd_groupnorm_workspace_bytes(). */
int vd_groupnorm_affine_f16(const void* x, const void* gamma, const void* beta, void* scale, void* shift, float* stats, int B,
                            int HW, int C, int groups, float eps, hipStream_t stream);

/* The gated feed-forward of a BasicTransformerBlock in one launch (inner width C = 320 only: vd_ff_geglu_supported):
 *     y[m] = res[m] + ( v (*) gelu_erf(g) ) W2^T + b2,     [v | g] = LayerNorm(x[m]) W1^T + b1
 * x, res, y: fp16 [M][C]; w1_packed: fp16 [8C][C] with LayerNorm's gamma folded in (W1 * gamma) and rows packed per 64 as
 * [32 value | 32 gate] (VD_ACT_GEGLU layout); b1_packed: fp16 [8C] = beta W1^T + b1, packed likewise; w2: fp16 [C][4C];
 * b2: fp16 [C].  The rows of x are layer-normalised in registers (biased variance, ln_eps), the [M, 4C] intermediate never
 * leaves the chip.  Replaces norm3 -> ff (GEGLU -> Linear) -> + x of lib/model_zoo/attention.py:37-64,214-218 and this
 * library's vd_row_stats_f16 -> vd_gemm_f16(VD_ACT_GEGLU | VD_EPI_LNFOLD) -> vd_gemm_f16(VD_EPI_RESIDUAL) chain. */
int vd_ff_geglu_f16(const void* x, const void* w1_packed, const void* b1_packed, const void* w2, const void* b2,
                    const void* res, void* y, int64_t M, int C, float ln_eps, hipStream_t stream);
int vd_ff_geglu_supported(int C);
/* The same feed-forward with the C x C projections on either side of it in the SAME launch (ABI 6, C = 320:
 * vd_ff_chain_supported) -- the row-local tail of a 64x64-level transformer block,
 *     x1  = a wo^T + bo + x                 (given a: CrossAttention.to_out + residual, attention.py:192-193,216)
 *     y   = x1 + FF(LayerNorm(x1))           (attention.py:37-64,217)
 *     out = alpha (y wp^T + bp) + res        (given wp: SpatialTransformer.proj_out + skip / context mixing, attention.py:262-266)
 * instead of vd_gemm_f16 -> vd_ff_geglu_f16 -> vd_gemm_f16.  Without `a` the feed-forward reads x itself; without `wp` it
 * stores y.  x1_scratch: [M][C] fp16 the kernel parks x1 in (needed with `a`).  out_stats (with wp, M % 128 == 0): fp32
 * [M / 128][C][2] per-channel (mean, M2) over blocks of 128 stored rows, the VdGemmDesc.out_stats format with R = 128. */
typedef struct VdFfChain {
    const void* x;          /* fp16 [M][C] */
    const void* a;          /* fp16 [M][C] or NULL */
    const void* wo;         /* fp16 [C][C] */
    const void* bo;         /* fp16 [C] */
    void* x1_scratch;       /* fp16 [M][C] */
    const void* w1_packed;  /* fp16 [8C][C]: LayerNorm-folded, GEGLU-packed (as vd_ff_geglu_f16) */
    const void* b1_packed;  /* fp16 [8C] */
    const void* w2;         /* fp16 [C][4C] */
    const void* b2;         /* fp16 [C] */
    const void* wp;         /* fp16 [C][C] or NULL */
    const void* bp;         /* fp16 [C] */
    const void* res;        /* fp16 [M][C]: residual of the last projection */
    void* out;              /* fp16 [M][C] */
    float* out_stats;       /* or NULL */
    void* stat_sums;        /* or NULL (with out_stats): int64 [M / stat_img_rows][C][2], as VdGemmDesc.stat_sums (ABI 7) */
    int64_t stat_img_rows;  /* rows of one image (a multiple of 128) for stat_sums */
    int64_t M;
    int32_t C;
    float ln_eps, alpha;
    int32_t reserved;
} VdFfChain;
int vd_ff_chain_f16(const VdFfChain* chain, hipStream_t stream);
int vd_ff_chain_supported(int C);

/* GroupNorm(groups) [+ SiLU] over channels-last input that may be the concatenation of two tensors.
 * stats is a caller-provided fp32 scratch of vd_groupnorm_workspace_bytes().
 * Replaces GroupNorm32 + SiLU (lib/model_zoo/diffusion_utils.py:175-191, openaimodel.py:196-200,230-237),
 * Normalize (attention.py:76-77, autokl_modules.py:38-39) and the torch.cat in vd.py:371. */
int vd_groupnorm_silu_f16(const void* x0, int c0, const void* x1, int c1, const void* gamma, const void* beta,
                          void* y, float* stats, int B, int HW, int groups, float eps, int apply_silu,
                          hipStream_t stream);
size_t vd_groupnorm_workspace_bytes(int B, int HW, int C, int groups);

/* ---- GroupNorm without its own statistics pass (ABI 5, csrc/gn_fused.hip) ----------------------------------------------
 * The producer of a tensor emits per-channel partial statistics while it stores it (VdGemmDesc.out_stats): fp32
 * [B * T][C][2] = (mean, M2 = sum (x - mean)^2) over T blocks of HW / T rows per sample.  Per channel, so the consumer folds
 * the channels of one tensor, or of both tensors of a skip concat (vd.py:371), into its groups whatever the producers' tile
 * shapes were, and a tensor consumed twice (every skip connection) is measured once.
 * vd_groupnorm_from_stats_f16: y = GroupNorm(cat(x0, x1)) [+ SiLU] in ONE launch that reads x once -- each block folds the
 *   partials of its groups (Chan's parallel variance update, fp32) and streams its panel.  Same contract as
 *   vd_groupnorm_silu_f16 otherwise; replaces GroupNorm32 + SiLU / Normalize at the same reference lines.
 * vd_chan_stats_f16: the same partials computed with one read of x [M][C] (row stride ldx), one per rows_per_partial
 *   consecutive rows (a divisor of the sample's row count): for tensors whose producer cannot emit them, and the reference
 *   the producers are tested against.
 * vd_gn_table_f32: partials -> the normalisation as a per-(sample, channel) affine map, fp32 table [B][2][C0 + C1]:
 *   scale = rstd * gamma, shift = beta - mean * scale;  vd_gn_apply_table_f16: y = act(x * scale + shift), elementwise. */
int vd_groupnorm_from_stats_f16(const void* x0, int c0, const float* stats0, int T0, const void* x1, int c1,
                                const float* stats1, int T1, const void* gamma, const void* beta, void* y, int B, int HW,
                                int groups, float eps, int apply_silu, hipStream_t stream);
int vd_chan_stats_f16(const void* x, long M, int C, int ldx, int rows_per_partial, float* stats, hipStream_t stream);
int vd_gn_table_f32(const float* stats0, int T0, int c0, const float* stats1, int T1, int c1, int B, int HW,
                    const void* gamma, const void* beta, int groups, float eps, float* table, hipStream_t stream);
int vd_gn_apply_table_f16(const void* x0, int c0, const void* x1, int c1, int B, int HW, const float* table, int apply_silu,
                          void* y, hipStream_t stream);
/* vd_gn_apply_sums_f16 (ABI 7): y = act(GroupNorm(cat(x0, x1))) from the per-(image, channel) fixed-point sums the producers
 * accumulated (VdGemmDesc.stat_sums: sums0 int64 [B][c0][2], sums1 int64 [B][c1][2] or NULL): every block folds the sums of its
 * image into (mean, rstd) per group in fp64 (8 lanes per group, no partial lists), builds scale / shift of its channel octets in
 * registers and streams its rows once.  groups <= 32.  Replaces the vd_gn_table_f32 + vd_gn_apply_table_f16 pair (same reference
 * lines as vd_groupnorm_silu_f16). */
int vd_gn_apply_sums_f16(const void* x0, int c0, const void* sums0, const void* x1, int c1, const void* sums1, int B, int HW,
                         const void* gamma, const void* beta, int groups, float eps, int apply_silu, void* y, hipStream_t stream);
/* The same map as fp16 [B][C0 + C1] scale / shift vectors: the gn_scale / gn_shift operands of vd_gemm_row320_chain_f16
 * (what vd_groupnorm_affine_f16 computes with a pass over x).  center (ABI 7, may be NULL): fp16 [B][C0 + C1] = fp16(group mean);
 * shift is then beta - (mean - center) * scale, for the centred application described at vd_gemm_row320_chain_f16. */
int vd_gn_affine_from_stats_f16(const float* stats0, int T0, int c0, const float* stats1, int T1, int c1, int B, int HW,
                                const void* gamma, const void* beta, int groups, float eps, void* scale, void* shift,
                                void* center, hipStream_t stream);

/* GroupNorm(groups) [+ SiLU] of the 0-D (text-latent) data flow: FCBlock normalises the flattened [C, sdim] vector of a
 * sample with one affine pair per flat element (reference openaimodel.py:2084-2141 with the [C, sdim, 1] -> C*sdim view
 * of FCBlock_MultiDim, :2295-2332).  x0 (++ x1 on channels): [B, S, C] channels-last; gamma / beta: [S, C] (the
 * reference's c * sdim + s order re-ordered once at load); y: [B, S, C]. */
int vd_groupnorm0d_silu_f16(const void* x0, int c0, const void* x1, int c1, const void* gamma, const void* beta, void* y,
                            int B, int S, int groups, float eps, int apply_silu, hipStream_t stream);

/* LayerNorm over the last dim of [rows][C]. Replaces nn.LayerNorm in BasicTransformerBlock
 * (lib/model_zoo/attention.py:205-207) and the HF CLIP layer norms. */
int vd_layernorm_f16(const void* x, const void* gamma, const void* beta, void* y, int rows, int C, float eps,
                     hipStream_t stream);
/* (mean, rstd = 1 / sqrt(var + eps)) of every row of x [rows][ldx >= C] (fp16, biased variance over the C columns,
 * two-pass in registers) -> stats fp32 [rows][2]: the statistics half of nn.LayerNorm for VD_EPI_LNFOLD. */
int vd_row_stats_f16(const void* x, float* stats, int64_t rows, int C, int64_t ldx, float eps, hipStream_t stream);

/* Fused softmax(Q K^T * scale) V, online softmax, no score tensor in HBM.
 * q [B][Nq][ldq], k/v [B][Nk][ldk/ldv], out [B][Nq][ldo]; head h occupies columns h*D..h*D+D-1.
 * D in {40, 64, 80, 160}.  causal = 1 masks key > query (CLIP text tower); causal = 1 + n masks key > query + n (n memory
 * slots in front of the keys that every query sees: the latent memory of the Optimus GPT-2 decoder, n = 1).
 * Replaces CrossAttention.forward einsum/softmax/einsum (lib/model_zoo/attention.py:176-191). */
int vd_attention_f16(const void* q, const void* k, const void* v, void* out, int B, int H, int Nq, int Nk, int D,
                     int ldq, int ldk, int ldv, int ldo, int64_t sq, int64_t sk, int64_t sv, int64_t so,
                     float scale, int causal, hipStream_t stream);

/* The query side of a cross-attention layer in one launch (vd_xattn_supported: head dim 40 / 80 / 160, width H*D a
 * multiple of 64):
 *     out[b, n, h*D..] = softmax( (LayerNorm(x[b, n]) Wq_h^T) K_h^T * scale ) V_h
 * x, out: fp16 [B][Nq][H*D] contiguous; wq: fp16 [H*D][H*D] with LayerNorm's gamma folded in, bq: fp16 [H*D] = Wq beta (or
 * NULL), colsum: fp32 [H*D] row sums of the folded wq (the VD_EPI_LNFOLD operands of vd_gemm_f16); k / v [B][Nk][ldk / ldv]
 * the pre-projected context, head h in columns h*D..h*D+D-1.  Row statistics (biased variance, ln_eps) are accumulated
 * inside the projection loop; the query tensor never exists in memory.  Replaces norm2 -> to_q -> einsum -> softmax ->
 * einsum of lib/model_zoo/attention.py:170-193,216 and this library's vd_row_stats_f16 -> vd_gemm_f16(VD_EPI_LNFOLD) ->
 * vd_attention_f16 chain. */
int vd_xattn_f16(const void* x, const void* wq, const void* bq, const float* colsum, float ln_eps, const void* k,
                 const void* v, void* out, int B, int H, int Nq, int Nk, int D, int ldk, int ldv, int64_t sk, int64_t sv,
                 float scale, hipStream_t stream);
int vd_xattn_supported(int H, int D);

/* Row softmax fp32 [rows][n] -> fp16 (VAE AttnBlock, lib/model_zoo/autokl_modules.py:192). */
int vd_softmax_rows_f32_f16(const float* s, void* p, int64_t rows, int n, hipStream_t stream);
/* softmax(scale * s) per row in fp32: token probabilities of the Optimus GPT-2 decoder, scale = 1 / temperature
 * (lib/model_zoo/optimus.py:679-681: logits / temperature -> softmax -> multinomial). */
int vd_softmax_rows_f32_f32(const float* s, float* p, int64_t rows, int n, float scale, hipStream_t stream);

/* Sinusoidal timestep embedding, fp32 math, fp16 out [B][dim] = [cos | sin].
 * Replaces timestep_embedding (lib/model_zoo/diffusion_utils.py:131-151). */
int vd_timestep_embedding_f16(const int64_t* t, void* out, int B, int dim, float max_period, hipStream_t stream);

/* Classifier-free-guidance combine + DDIM update in one pass (fp32 math):
 *   e = e_u + s (e_c - e_u); pred_x0 = (x - sqrt(1-a_t) e)/sqrt(a_t);
 *   x_prev = sqrt(a_prev) pred_x0 + sqrt(1 - a_prev - sigma^2) e + sigma * noise
 * eps holds [e_uncond ; e_cond] (2n elements) when guided != 0, else n elements.
 * Replaces p_sample_ddim (lib/model_zoo/ddim.py:144-170). */
int vd_cfg_ddim_step_f16(const void* x, const void* eps, const void* noise, void* x_prev, void* pred_x0, int64_t n,
                         int guided, float guidance_scale, float a_t, float a_prev, float sigma,
                         float sqrt_one_minus_at, hipStream_t stream);

/* Same update with the step scalars in device memory: coef[6] = {guidance scale, 1/sqrt(a_t), sqrt(a_prev),
 * sqrt(1 - a_prev - sigma^2), sigma, sqrt(1 - a_t)}.  Lets one captured HIP graph of a DDIM step be replayed for
 * every step (the host only refreshes coef / the timestep tensor between replays). */
int vd_cfg_ddim_step_dev_f16(const void* x, const void* eps, const void* noise, void* x_prev, void* pred_x0, int64_t n,
                             int guided, const float* coef, hipStream_t stream);

/* q_sample: out = sa[b] * x0 + sb[b] * noise  (lib/model_zoo/vd.py:221-224). */
int vd_q_sample_f16(const void* x0, const void* noise, const float* sa, const float* sb, void* out, int B,
                    int64_t per_batch, hipStream_t stream);

/* layout changes at the API boundary (reference tensors are NCHW) */
int vd_nchw_to_nhwc_f16(const void* x, void* y, int B, int C, int H, int W, hipStream_t stream);
int vd_nhwc_to_nchw_f16(const void* x, void* y, int B, int C, int H, int W, float scale, float shift, int clamp01,
                        hipStream_t stream);

/* Small-Cin im2col: A[m][kpad] (zero padded, k = (ky*ks+kx)*C + c) from a strided fp16 image, with the
 * affine x*in_scale+in_shift applied to valid pixels (fuses `x*2-1`, autokl.py:34 and `z/scale`, vd.py:294). */
int vd_im2col_small_f16(const void* x, void* a, int B, int C, int Hin, int Win, int Hout, int Wout, int ksize,
                        int stride, int pad, int64_t sb, int64_t sc, int64_t sy, int64_t sx, int kpad,
                        float in_scale, float in_shift, hipStream_t stream);

/* DiagonalGaussianDistribution.sample fused with the latent scale:
 *   z = (mean + exp(0.5*clamp(logvar,-30,20)) * noise) * scale ; moments channels-last [M][2*zc], z NCHW
 * (lib/model_zoo/distributions.py:24-37, vd.py:282-289). */
int vd_diag_gaussian_sample_f16(const void* moments, const void* noise, void* z, int B, int zc, int HW, float scale,
                                hipStream_t stream);

/* out = a * x + b * y (context mixing helpers, vd.py:383-396) */
int vd_axpby_f16(const void* x, const void* y, void* out, float a, float b, int64_t n, hipStream_t stream);

/* out = f(x) element-wise (in place allowed): erf-GELU of BERT's intermediate layer and the tanh of its pooler in the
 * Optimus encoder (lib/model_zoo/optimus_models/optimus_bert.py:139-150 `gelu`, :451-463 BertPooler). */
#define VD_UNARY_GELU_ERF 0
#define VD_UNARY_TANH 1
int vd_unary_f16(const void* x, void* out, int op, int64_t n, hipStream_t stream);

/* CLIP helpers (arithmetic of HF transformers CLIPModel as called from lib/model_zoo/clip.py) */
int vd_embed_tokens_f16(const int64_t* ids, const void* tok_emb, const void* pos_emb, void* out, int B, int L, int C,
                        hipStream_t stream);
int vd_clip_vision_embed_f16(const void* patches, const void* class_emb, const void* pos_emb, const float* token_scale,
                             void* out, int B, int L, int C, hipStream_t stream);
int vd_patchify_f16(const void* pixels, void* a, int B, int C, int H, int W, int P, int kpad, hipStream_t stream);
/* z[b][l][:] *= row_scale[b][l] / ||ref[b][:]||  (ref row = z[b][pool_idx[b]], or `ref` if given) */
int vd_scale_by_row_norm_f16(void* z, const void* ref, const int32_t* pool_idx, const float* row_scale, int B, int L,
                             int C, hipStream_t stream);

/* CLIP image pre-processing on the device, bit-exact with the reference's host path (lib/model_zoo/clip.py:88-94:
 * torchvision ToPILImage -> HuggingFace CLIPProcessor = Pillow 8-bit bicubic resize of the shortest edge, centre crop,
 * rescale, normalise).  img [B,3,H,W]: img_kind 0 = float32 / 1 = float16 in [0,1] (quantised like ToPILImage:
 * mul(255).byte()), 2 = uint8.  (rh, rw) = resized size, crop window (crop_t, crop_l, size).  hb/hk, vb/vk: Pillow's
 * fixed-point taps per output column / row (device int32: bounds[out][2] = {first input index, taps}, kk[out][ks], 22
 * fractional bits; ks = 0 and null tables when that axis is not resized).  norm_table: device float32 [256][3] =
 * (level * (1/255) - mean[c]) / std[c].  tmp: B*3*H*size bytes of scratch.  out [B,3,size,size] float16. */
int vd_clip_preprocess_f16(const void* img, int img_kind, int B, int H, int W, int rh, int rw, const int32_t* hb,
                           const int32_t* hk, int hks, const int32_t* vb, const int32_t* vk, int vks, int crop_t, int crop_l,
                           int size, const float* norm_table, uint8_t* tmp, void* out, hipStream_t stream);

/* Decoded image [B,3,H,W] (img_kind 0 = float32, 1 = float16, values in [0,1]) -> uint8 [B,H,W,3], the arithmetic of
 * torchvision ToPILImage on the output of vae_decode (reference app.py:319): mul(255) in the tensor's dtype, .byte(). */
int vd_image_to_u8(const void* img, int img_kind, int B, int H, int W, uint8_t* out, hipStream_t stream);

/* Mask -> per-token weights of the masked CLIP image context (lib/model_zoo/clip.py:104-122): masks [B,1,H,W]
 * (mask_kind 0 = float32, 1 = float16) are clamped to [0,1], resized to size x size like F.interpolate(mode='bilinear')
 * and averaged per patch x patch cell; out [B][1 + (size/patch)^2] fp32 = [global mean | patch means]. */
int vd_mask_patch_weights(const void* masks, int mask_kind, int B, int H, int W, int size, int patch, float* out,
                          hipStream_t stream);

/* 'Simple' colour adjustment of image variation (app.py:373-379): per image and channel
 *   out = clamp((img - mean(img)) / std(img) * std(ref) + mean(ref), 0, 1)   (unbiased std over the H*W pixels)
 * img / out [B,3,H,W] fp16, ref [3,H,W] (ref_batch_stride = 0: one input image for the whole batch, as the app does) or
 * [B,3,H,W] (ref_batch_stride = 3*H*W). */
int vd_color_adjust_f16(const void* img, const void* ref, void* out, int B, int H, int W, int64_t ref_batch_stride,
                        hipStream_t stream);

/* Focus control of the image context ("adjust_rank", app.py:48-127): per sample x [L][C] (fp16 in, fp16 out)
 *   A = x - rowmean(x);  U = top-q left singular vectors of A (= what torch.pca_lowrank(q, niter=100) converges to)
 *   x_new = keep * A + sum_i g[i] * U[:,i] (U[:,i]^T A) + rowmean(x);   y = x_new * std(x) / std(x_new)   (unbiased std)
 * with g[i] = f_i - keep from the reference's level -> singular-value-scale curves (host side, lib/app_ops.py); keep = 1
 * keeps the remainder beyond rank q (lvl < 0.5), keep = 0 drops it (lvl > 0.5).  U comes from `iters` steps of a
 * 32-column subspace iteration on A A^T in fp32.  32 <= L <= 512, q <= 32.  ws: vd_adjust_rank_workspace_bytes(). */
int vd_adjust_rank_f16(const void* x, void* y, int B, int L, int C, int q, const float* g, float keep, int iters, float* ws,
                       hipStream_t stream);
size_t vd_adjust_rank_workspace_bytes(int B, int L, int C, int q);

/* diagnostics */
const char* vd_last_error(void);
int vd_abi_version(void);
/* writes lane->(row,col) maps of the 32x32x16 f16 MFMA as observed on the device; used by tests */
int vd_probe_mfma_layout(int32_t* out_a_k, int32_t* out_c_row, int32_t* out_c_col, hipStream_t stream);
/* ds_read_b64_tr_b16 as observed on the device: LDS holds lds[i] = i (4096 int16); lane l reads at byte address
 * addr_bytes[l] (64 entries) and out[l*4 + j] receives its four 16-bit results; used by tests (the attention kernel's
 * V operand relies on this gather) */
int vd_probe_lds_tr16(const int32_t* addr_bytes, int16_t* out, hipStream_t stream);
/* XCD (HW_REG_XCC_ID) of every block of a grid_x x grid_y launch of 256-thread blocks, int32 [grid_y][grid_x]: pins the
 * dispatcher's round-robin placement (block b of the linearised grid runs on XCD b % 8) that the tile order of the GEMM /
 * convolution kernels exploits for L2 locality and the ticketed split of conv3x3_halo_kernel relies on. */
int vd_probe_xcc_ids(int32_t* out, int grid_x, int grid_y, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VD_HIP_H */
