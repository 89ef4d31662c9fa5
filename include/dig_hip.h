/* dig_hip.h - C ABI of libdig_hip.so: the MI355X (gfx950) kernels under the DiG diffusion-guidance
 * hot path.  The reference (yoon307/DiG) has no FFI layer - the path sits behind nn.Module calls that
 * bottom out in ATen ops - so each entry point below names the reference call site(s) whose ATen op(s)
 * it replaces ("ddp.py" = denoising_diffusion_pytorch/denoising_diffusion_pytorch.py).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; nothing here allocates, frees or syncs;
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - return 0 on success, a negative DIG_E* code otherwise; dig_last_error() gives the message;
 *   - activations are NHWC ("pixel-major"): [B][H][W][C], C contiguous; `dtype` selects the storage
 *     type of activations and weights (DIG_BF16 or DIG_F32); accumulation/statistics are fp32;
 *   - channel counts must be multiples of 8 (16-byte vectors); buffers must be 16-byte aligned.
 */
#ifndef DIG_HIP_H
#define DIG_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define DIG_OK 0
#define DIG_EINVAL (-1)  /* bad shape / argument */
#define DIG_EALIGN (-2)  /* misaligned pointer or stride */
#define DIG_ELAUNCH (-3) /* hipGetLastError() after launch */

#define DIG_BF16 0
#define DIG_F32 1

int dig_version(void);
const char* dig_last_error(void);

/* ---- conv3x3, implicit GEMM on MFMA with an LDS-staged halo tile ------------------------------
 * Replaces: Block.proj (ddp.py:154,159), Upsample's nearest x2 + conv (ddp.py:90-94), the last-stage
 * 3x3 convs (ddp.py:349,366), torch.cat of skip connections (ddp.py:423,428,441) and - through the
 * input prologue - GroupNorm + scale/shift + SiLU of the PREVIOUS Block (ddp.py:160-166).
 *   in0/in1 : the two halves of a channel concat [B][Hi][Wi][C0|C1] (in1 may be NULL, C1 = 0);
 *             with up2 != 0 the inputs are [B][H/2][W/2][C] and are read with nearest-x2 addressing.
 *   aff0    : NULL, or [B][C0][2] fp32 (a, b): in0 is read as silu(a*x + b)  (GroupNorm-apply prologue)
 *   w       : packed [ceil(Cout/BN)][chunks][9 taps][BN][KC] - the per-chunk LDS weight slabs, contiguous, zero padded;
 *             chunks = ceil(C0/KC) + ceil(C1/KC), (BN, KC) from dig_conv3x3_config(Cout, dtype);   bias: [Cout] fp32
 *   out     : [B][H][W][Cout] raw conv output (+bias)
 *   stats   : NULL, or [B][groups][tiles][2] fp32 per-tile (sum, sumsq) of out over each GroupNorm group,
 *             tiles = dig_conv3x3_tiles(H, W, C0, C1, Cout, groups, dtype); reduced by dig_gn_finalize (deterministic, no atomics). */
typedef struct {
    const void* in0;
    const void* in1;
    const float* aff0;
    const void* w;
    const float* bias;
    void* out;
    float* stats;
    long long in0_bstride; /* elements between images of in0; 0 = dense (Hi*Wi*C0) */
    int B, H, W, C0, C1, Cout, groups, up2, dtype;
} dig_conv3x3_args;
int dig_conv3x3_tiles(int H, int W, int C0, int C1, int Cout, int groups, int dtype);
/* Layout of the packed weight slabs for this layer: 0 = plain [n_tile][chunk][9 taps][BN][KC]; 2 = the LDS image of the
 * double-buffered wide-layer kernel (conv3x3_db.hip), which copies the slabs global -> LDS without touching them:
 * per 128-channel N tile and per PAIR of 16-channel chunks  main(2P) | main(2P+1) | w8(P), K = 32 of one
 * v_mfma_f32_16x16x32_bf16 = two taps x 16 channels, rows permuted so that a lane's accumulators are 8 consecutive
 * output channels, 16-byte slots XOR-swizzled for conflict-free ds_read_b128 (exact definition: dig_amd/ops.py:_pack_pairs;
 * 1 was the 32x32x16 image of rounds 1-2). */
int dig_conv3x3_wlayout(int C0, int C1, int Cout, int dtype);
int dig_conv3x3_config(int Cout, int dtype, int* BN, int* KC);
int dig_conv3x3(const dig_conv3x3_args* a, void* stream);

/* ---- batched "NT" GEMM on MFMA: C[z][m][n] = epi( alpha * sum_k A[z][m][k] * Bm[z][n][k] ) -------
 * Both operands are K-contiguous.  Replaces every conv1x1 / nn.Linear / einsum / matmul on the path:
 * to_qkv, to_out, res_conv, Downsample's Rearrange + conv1x1 (ddp.py:102-106,180,209-213,250-251),
 * final_conv (ddp.py:373), the time MLPs (ddp.py:172-175,318-320), Attend's two einsums
 * (attend.py:102,111), LinearAttention's second einsum (ddp.py:231), and the ViT linears / patch embed.
 *   A rows may be a concat of two tensors (a0: K0 columns, a1: K - K0 columns), or - a_mode 1 - the
 *   2x2 pixel-unshuffle gather of an NHWC tensor (row m = output pixel, K = 4*C ordered (p1,p2,c)).
 *   Batch index z = zo*z_inner + zi; every operand has an (outer, inner) element stride pair.
 *   Epilogue, in order: *alpha, +bias[n], +silu(gaff[img][n][0]*gsrc[m][n] + gaff[img][n][1]) with
 *   img = m / rows_per_img (GroupNorm-apply of a raw conv output fused into res_conv, ddp.py:160-166,194),
 *   GELU(erf) if gelu, +res[m][n]; store as dtype (or fp32 if out_f32) row-major [m*ldc+n], or - c_mode 1 -
 *   NCHW: out[(m/hw)*N*hw + n*hw + m%hw]. */
typedef struct {
    const void* a0;
    const void* a1;
    const void* b;
    void* c;
    const float* bias;
    const void* res;
    const void* gsrc;
    const float* gaff;
    long long sa_o, sa_i, sa1_o, sa1_i, sb_o, sb_i, sc_o, sc_i, sr_o, sr_i;
    int M, N, K, K0, lda0, lda1, ldb, ldc, ldr, ldg;
    int batch, z_inner;
    int a_mode, unsh_Ho, unsh_Wo, unsh_C;
    int c_mode, hw, rows_per_img;
    int gelu, out_f32, dtype;
    float alpha;
    float gsm_scale; /* != 0: after bias, softmax over every aligned group of 32 columns, times gsm_scale (q.softmax(dim=-2)*scale, ddp.py:224,227) */
} dig_gemm_args;
int dig_gemm_nt(const dig_gemm_args* a, void* stream);

/* ---- GroupNorm finalize: per-tile partials -> per-(image, channel) affine -----------------------
 * Replaces nn.GroupNorm statistics (ddp.py:155,160) + `x*(scale+1)+shift` folding (ddp.py:162-164):
 *   a[b][c] = rstd*gamma[c]*(1+scale[b][c]),  bb[b][c] = (beta[c]-mean*rstd*gamma[c])*(1+scale[b][c]) + shift[b][c]
 * ss: NULL or rows [b*ss_ld + ss_off .. ] holding (scale[0..C), shift[0..C)) = the time-MLP output chunk. */
typedef struct {
    const float* stats;
    const float* gamma;
    const float* beta;
    const float* ss;
    float* aff;
    int B, C, groups, tiles, hw, ss_ld, ss_off;
    float eps;
} dig_gn_finalize_args;
int dig_gn_finalize(const dig_gn_finalize_args* a, void* stream);

/* out = silu(a*x + b) [+ res]   (GroupNorm apply + SiLU + residual add; ddp.py:160-166,194 with identity res_conv) */
typedef struct {
    const void* x;
    const float* aff;
    const void* res;
    void* out;
    int B, HW, C, dtype;
} dig_gn_silu_add_args;
int dig_gn_silu_add(const dig_gn_silu_add_args* a, void* stream);

/* dig_gn_finalize + dig_gn_silu_add in ONE launch, for layers whose statistics are few partials per image (tiles <= 64): every
 * block first reduces the partials of its image (the same arithmetic, in the same order, as dig_gn_finalize: the affine is
 * bit-identical) into LDS, then applies out = silu(a*x + b) [+ res] to its slice of that image.  out may alias x.
 * Replaces nn.GroupNorm + scale/shift + SiLU (+ the identity-shortcut add) of ddp.py:155-166,194 without the affine table's
 * round trip through memory and without the 6 us launch that computed it. */
typedef struct {
    const void* x;
    const void* res;
    void* out;
    const float* stats;
    const float* gamma;
    const float* beta;
    const float* ss;
    int B, HW, C, dtype, groups, tiles, hw, ss_ld, ss_off;
    float eps;
} dig_gn_fin_silu_add_args;
int dig_gn_fin_silu_add(const dig_gn_fin_silu_add_args* a, void* stream);
/* largest `tiles` dig_gn_fin_silu_add accepts (above it the per-block reduction costs more than the separate launch) */
int dig_gn_fin_max_tiles(void);

/* ResnetBlock tail with a channel-changing shortcut (ddp.py:190-194), streaming kernel for small K and N (the 256x256
 * and 128x128 levels):  out[m][:] = [x0 | x1][m][:] @ w^T + bias + SiLU(gaff.a * gsrc[m][:] + gaff.b).
 *   x0 [M][K0], x1 [M][K1] (or NULL), w [N][K0+K1], gsrc [M][N], gaff [M / rows_per_img][N][2], out [M][N].
 * dig_pw_res_supported tells whether (K0, K1, N, dtype) is handled; otherwise use dig_gemm_nt with the same epilogue. */
typedef struct {
    const void* x0;
    const void* x1;
    const void* w;
    const float* bias;
    const void* gsrc;
    const float* gaff;
    void* out;
    long long M;
    int K0, K1, N, rows_per_img, dtype;
    /* Optional fused final_conv (ddp.py:373,443-448; K0+K1 = 128, N = 64 only): when fout != NULL the block output is not
     * stored (out may be NULL); instead fout[img][o][pixel] = fb[o] + sum_c fw[o][c] * result[pixel][c] in fp32, NCHW,
     * o < f_no <= 4, pixel < rows_per_img. */
    const float* fw;
    const float* fb;
    float* fout;
    int f_no;
} dig_pw_res_args;
int dig_pw_res_supported(int K0, int K1, int N, int dtype);
int dig_pw_res(const dig_pw_res_args* a, void* stream);

/* RMSNorm over channels per pixel (ddp.py:108-114): out = x / max(||x||,1e-12) * g * sqrt(C) [+ res] */
typedef struct {
    const void* x;
    const float* g;
    const void* res;
    void* out;
    long long rows;
    int C, dtype;
} dig_rmsnorm_args;
int dig_rmsnorm(const dig_rmsnorm_args* a, void* stream);

/* softmax over each contiguous row; in fp32 [rows][n] -> out dtype [rows][n]  (attend.py:106; ddp.py:225 uses
 * dig_row_max + dig_la_context instead).  in_f32: source rows are fp32 regardless of dtype. */
typedef struct {
    const void* in;
    void* out;
    long long rows;
    int n, in_f32, dtype;
} dig_row_softmax_args;
int dig_row_softmax(const dig_row_softmax_args* a, void* stream);

/* Full softmax attention with dim_head 32 (Attend.forward, attend.py:98-113, as called from ddp.py:254-262), flash
 * style: out[b][i][h*32+e] = sum_j softmax_j(scale * q_i.k_j) v[j][e]; the [n][n] logits stay in registers.
 *   qk : [B][n][ldqk] pixel-major, q of head h at columns h*32.., k at heads*32 + h*32..
 *   vT : [B][heads*32][n] channel-major;  out: [B][n][heads*32].  n % 8 == 0. */
typedef struct {
    const void* qk;
    const void* vT;
    void* out;
    int B, n, heads, ldqk, dtype;
    float scale;
} dig_flash_attn_args;
int dig_flash_attn(const dig_flash_attn_args* a, void* stream);

/* q.softmax(dim=-2) * scale of LinearAttention (ddp.py:224,227) on pixel-major q [rows][groups*32], in place */
typedef struct {
    void* q;
    long long rows;
    int groups, dtype;
    float scale;
} dig_group_softmax32_args;
int dig_group_softmax32(const dig_group_softmax32_args* a, void* stream);

/* LinearAttention context (ddp.py:225,229): k,v channel-major [32][n] slabs per (image, head); bh = b*heads + h.
 *   dig_row_max:   kmax[r] = max_n k[r][n]
 *   dig_la_context: per split s of n: part[bh][s][e][d] = sum_n v[e][n]*exp(k[d][n]-kmax[d]),  ksum[bh][s][d] = sum_n exp(..)
 *   dig_la_finalize: ctxT[bh][e][d] = scale * sum_s part / sum_s ksum   (stored as dtype, the B operand of the out GEMM) */
typedef struct {
    const void* in;
    float* out;
    long long rows;        /* total rows reduced; row i lives at (i / group_rows)*group_stride + (i % group_rows)*n */
    long long group_stride;
    int group_rows, n, dtype;
} dig_row_max_args;
int dig_row_max(const dig_row_max_args* a, void* stream);
typedef struct {
    const void* k;
    const void* v;
    const float* kmax;
    float* part;
    float* ksum;
    long long b_stride_k, h_stride_k, b_stride_v, h_stride_v; /* element strides of the (image, head) [32][n] slabs */
    int B, heads, n, splits, dtype;
} dig_la_context_args;
int dig_la_context(const dig_la_context_args* a, void* stream);
typedef struct {
    const float* part;
    const float* ksum;
    void* ctxT;
    int BH, splits, dtype;
    int transposed; /* 0: ctxT[bh][e][d]; 1: ctx[bh][d][e] */
    float scale;
} dig_la_finalize_args;
int dig_la_finalize(const dig_la_finalize_args* a, void* stream);

/* ---- fused LinearAttention (ddp.py:196-233): x is read twice and the result written once ------------------------
 *   dig_la_kvctx: per (image, pixel split), one wave per head: k,v = W'kv x with RMSNorm folded (g into the weights,
 *                 1/||x|| as a row scale), online softmax of k over pixels, ctxT[e][d] += v^T exp(k - m) with both MFMA
 *                 operands taken from the accumulators.  wkv: [256][C] = (k rows | v rows) * g.  Outputs per
 *                 (b, split, head): part [32 e][32 d] fp32, pmax[32 d], psum[32 d].  Needs 4 heads of 32, C in {16..256}.
 *   dig_la_merge: log-sum-exp merge of the splits -> ctx[b][h][d][e] in dtype (B operand of the M_b GEMM), and optionally
 *                 M_b itself (second einsum folded into to_out, ddp.py:231-232) in the same launch.
 *   dig_la_out  : q = softmax_d(W'q x) * scale, y = M_b q + bias, out = RMSNorm(y) * g2 + x.   mb: [B][dim][128]. */
typedef struct {
    const void* x;
    const void* wkv;
    float* part;
    float* pmax;
    float* psum;
    int B, n, C, hid, splits, dtype;
} dig_la_kvctx_args;
int dig_la_kvctx(const dig_la_kvctx_args* a, void* stream);
typedef struct {
    const float* part;
    const float* pmax;
    const float* psum;
    void* ctx;
    int B, splits, dtype;
    /* Optional (mb != NULL): also form M_b[b][c][h*32+d] = sum_e wo[c][h*32+e] * ctx[b][h][d][e] in the same launch
     * (wo: to_out weight [dim][128] in dtype, mb: [B][dim][128] in dtype); ctx may then be NULL. */
    const void* wo;
    void* mb;
    int dim;
} dig_la_merge_args;
int dig_la_merge(const dig_la_merge_args* a, void* stream);
typedef struct {
    const void* x;
    const void* wq;
    const void* mb;
    const float* bias;
    const float* g2;
    void* out;
    int B, n, C, dim, dtype;
    float scale;
} dig_la_out_args;
int dig_la_out(const dig_la_out_args* a, void* stream);

/* ---- allocation-free LinearAttention context: dig_la_kvctx followed by dig_la_merge, without a caller workspace ----------
 * NO COUNTERPART IN THE REFERENCE (it has no workspace notion at all).  It exists for callers that cannot allocate device
 * memory around the call: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  The split
 * partials that dig_la_kvctx takes as (part, pmax, psum) live in a static device array inside the library instead
 * (17.3 MB per GPU, held until the process ends); the caller passes only the input, the weights and the output.
 *   x [B][n][C], wkv [256][C] (as dig_la_kvctx)  ->  ctx [B*4][32 d][32 e] in dtype (as dig_la_merge).
 * Limit: B * splits <= dig_la_ctx_noalloc_max_slots() (1024), else DIG_EINVAL with a message.  Results are bit-identical
 * to dig_la_kvctx + dig_la_merge with the same `splits`.  Calls no hipMalloc / hipFree / allocating routine and does not
 * synchronise.  All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two threads) race. */
typedef struct {
    const void* x;
    const void* wkv;
    void* ctx;
    int B, n, C, hid, splits, dtype;
} dig_la_ctx_noalloc_args;
int dig_la_ctx_noalloc_max_slots(void);
int dig_la_ctx_noalloc(const dig_la_ctx_noalloc_args* a, void* stream);

/* ---- stem: normalisation + q_sample + conv7x7 -------------------------------------------------------
 * Replaces denormforDiff/normalize (model.py:499-508, ddp.py:82-83), extract + q_sample (ddp.py:452-455,
 * 853-860) and init_conv (ddp.py:296,388).  img, noise: NCHW fp32 [B][3][H][W];
 *   x0 = pre_scale[c]*img + pre_shift[c];  x_t = ca[t[b]]*x0 + cs[t[b]]*noise  (written to x_t, NCHW fp32, if non-NULL)
 *   out = conv7x7(x_t) + bias  as NHWC dtype [B][H][W][Cout] (Cout % 8 == 0);
 *   w: [Cout][160] in dtype, k = (ky*7+kx)*3 + c for k < 147, zero for 147 <= k < 160 (the kernel's LDS weight panel). */
typedef struct {
    const float* img;
    const float* noise;
    const long long* t;
    const float* sqrt_ac;
    const float* sqrt_1mac;
    const void* w;
    const float* bias;
    float* x_t;
    void* out;
    float pre_scale[3], pre_shift[3];
    int B, H, W, Cout, dtype;
} dig_stem_args;
int dig_qsample_stem(const dig_stem_args* a, void* stream);

/* out = post_scale[c]*(ca[t[b]]*x + sign*cs[t[b]]*y) + post_shift[c], NCHW fp32 (q_sample ddp.py:853-860 with sign=+1;
 * predict_start_from_v ddp.py:630-634 with sign=-1, fused with unnormalize + normforCls, model.py:305-310,510-519) */
typedef struct {
    const float* x;
    const float* y;
    const long long* t;
    const float* sqrt_ac;
    const float* sqrt_1mac;
    float* out;
    float post_scale[3], post_shift[3];
    float sign;
    int B, HW;
} dig_sched_axpby_args;
int dig_sched_axpby(const dig_sched_axpby_args* a, void* stream);

/* ---- heads and forward losses around the MCTformer passes of model_WSSS.update (models/model_ECCV24_DiG.py:324-421) ---- */

/* in: NHWC fp32 head map [B][P][C].  relu_nchw (nullable) [B][C][P] = relu(in) transposed (outputs['cams'] / ['cam_diff'],
 * mctformer.py:298,336-339); pooled (nullable) [B][C] = mean(topk(relu(v),k)) - mean(topk(relu(-v),k)), k = min(3,P)
 * (the patch logits 'pcls' / 'diff_pcls', mctformer.py:300,341). */
typedef struct {
    const float* in;
    float* relu_nchw;
    float* pooled;
    int B, P, C;
} dig_head_pool_args;
int dig_head_pool(const dig_head_pool_args* a, void* stream);

/* out[b][c] = mean_d( x1 ? (x0[b][c][d] + x1[b][c][d]) / 2 : x0[b][c][d] ); image b of x0/x1 starts at b*bstride floats
 * (the class-token logits 'cls' = x[:, :C].mean(-1), mctformer.py:295, and 'diff_cls' of the two LFCA outputs, :287-288) */
typedef struct {
    const float* x0;
    const float* x1;
    float* out;
    long long bstride0, bstride1;
    int B, C, D;
} dig_row_mean_args;
int dig_row_mean(const dig_row_mean_args* a, void* stream);

/* out[0] (+)= scale * F.multilabel_soft_margin_loss(x, y), x / y [B][C] fp32 (model.py:337-340,358-359,406-409) */
typedef struct {
    const float* x;
    const float* y;
    float* out;
    float scale;
    int accumulate, B, C;
} dig_msm_loss_args;
int dig_msm_loss(const dig_msm_loss_args* a, void* stream);

/* out[0] (+)= scale * KLDivLoss(reduction='batchmean', log_target=True)(log cat(S, 1-S), log cat(Tt, 1-Tt)) with
 * S = softmax(student / T, 1), Tt = softmax(teacher / T, 1), [B][C] fp32 (model.py:347-355,360-361) */
typedef struct {
    const float* student;
    const float* teacher;
    float* out;
    float T, scale;
    int accumulate, B, C;
} dig_kd_bin_loss_args;
int dig_kd_bin_loss(const dig_kd_bin_loss_args* a, void* stream);

/* partial[b*C+c] = sum_hw | (max_norm(a)[b][c] * (a_times_label ? label : 1) - max_norm(resize(b))[b][c]) * label[b][c] |
 * a [B][C][ha][wa], b [B][C][hb][wb] bilinearly resized (align_corners=False) to ha x wa, max_norm = model.py:572-578
 * (loss_cam_consistency model.py:368-390 with a = cam_diff, b = cams; loss_rcams_consistency :414-418 with a = rcams) */
typedef struct {
    const float* a;
    const float* b;
    const float* label;
    float* partial;
    int B, C, ha, wa, hb, wb, a_times_label;
} dig_maxnorm_l1_args;
int dig_maxnorm_l1(const dig_maxnorm_l1_args* a, void* stream);

/* out[0] (+)= scale * sum(in[0..n))  (the .mean() over the partials above; the weighted total loss_trm, model.py:341-419) */
typedef struct {
    const float* in;
    float* out;
    float scale;
    int accumulate, n;
} dig_sum_scale_args;
int dig_sum_scale(const dig_sum_scale_args* a, void* stream);

/* One DDIM update of ddim_sample_from_t (ddp.py:792-814) for objective 'pred_v', elementwise over n fp32 values:
 *   x0  = clamp(sqrt_ac*x - sqrt_1mac*v, -1, 1)                      model_predictions, clip_x_start=True (ddp.py:662-666,630-634)
 *   eps = (sqrt_recip*x - x0) / sqrt_recipm1                         predict_noise_from_start (ddp.py:618-622)
 *   out = last ? x0 : x0*sqrt_alpha_next + c*eps + sigma*noise       (ddp.py:797-812; noise may be NULL when sigma == 0)
 * every product, sum and the division rounded separately, as the reference's tensor expression evaluates them.
 * The schedule coefficients are the caller's host scalars for this step's (time, time_next).
 * out_unnorm (nullable) also receives unnormalize(out) = (out + 1) * 0.5 (ddp.py:818, normalize_to_neg_one_to_one's inverse). */
typedef struct {
    const float* x;
    const float* v;
    const float* noise;
    float* out;
    float* out_unnorm;
    float sqrt_ac, sqrt_1mac, sqrt_recip, sqrt_recipm1, sqrt_alpha_next, c, sigma;
    int last;
    long long n;
} dig_ddim_step_args;
int dig_ddim_step(const dig_ddim_step_args* a, void* stream);

/* SinusoidalPosEmb (ddp.py:118-130): out[b][0..dim) fp32 = [sin(t*w_i), cos(t*w_i)] stored as dtype */
typedef struct {
    const long long* t;
    void* out;
    int B, dim, dtype;
} dig_sinusoid_args;
int dig_sinusoid(const dig_sinusoid_args* a, void* stream);

/* Time embedding head in one launch: SinusoidalPosEmb(dim) -> Linear(dim, tdim) -> GELU(erf) -> Linear(tdim, tdim)
 * (ddp.py:118-130,316-321; Unet.forward ddp.py:391), fp32.  temb [B][tdim] = the embedding, st [B][tdim] = SiLU(temb)
 * (first half of every ResnetBlock's `mlp`, ddp.py:172-173).  w1 [tdim][dim], w3 [tdim][tdim] row-major. */
typedef struct {
    const long long* t;
    const float* w1;
    const float* b1;
    const float* w3;
    const float* b3;
    float* temb;
    float* st;
    int B, dim, tdim;
} dig_time_embed_args;
int dig_time_embed(const dig_time_embed_args* a, void* stream);

/* Every ResnetBlock's Linear(tdim, 2*C_out) (ddp.py:174,186-188) as one skinny fp32 GEMM: out[b][n] = bias[n] + st[b][:] . w[n][:],
 * st [B][K], w [N][K] (the 19 projections concatenated along N), out [B][N]; exact fp32 MFMA. */
typedef struct {
    const float* st;
    const float* w;
    const float* bias;
    float* out;
    int B, N, K;
} dig_time_mlp_args;
int dig_time_mlp(const dig_time_mlp_args* a, void* stream);

/* elementwise activation on a flat buffer: act 0 = SiLU (ddp.py:173), 1 = GELU(erf) (ddp.py:319) */
typedef struct {
    const void* in;
    void* out;
    long long n;
    int act, in_f32, out_f32, dtype;
} dig_act_args;
int dig_act(const dig_act_args* a, void* stream);

/* bilinear resize, align_corners=True, NHWC dtype [B][H][W][C] -> NCHW fp32 slice of a [B][Ctot][h][w] tensor at
 * channel offset c_off (F.interpolate ddp.py:418,434 fused with torch.cat(feat,1), model.py:313) */
typedef struct {
    const void* in;
    float* out;
    int B, H, W, C, h, w, Ctot, c_off, dtype;
} dig_bilinear_ac_args;
int dig_bilinear_ac_concat(const dig_bilinear_ac_args* a, void* stream);

/* layout converters (test / boundary glue): NCHW fp32 <-> NHWC dtype.  out_ld (nhwc_to_nchw only; 0 = HW): pitch in floats
 * of the output's [B][C] rows, >= HW - the columns HW..out_ld-1 are left untouched (LFCA pads v^T rows to the GEMM's K vector) */
typedef struct {
    const void* in;
    void* out;
    int B, C, HW, dtype, out_ld;
} dig_layout_args;
int dig_nchw_f32_to_nhwc(const dig_layout_args* a, void* stream);
int dig_nhwc_to_nchw_f32(const dig_layout_args* a, void* stream);

/* ======================= MCTformer class-token attention -> CAM path (fp32 throughout) ======================= */

/* PatchEmbed im2row (networks/vision_transformer.py:238-245): img NCHW fp32 [B][3][H][W] -> rows [B*hp*wp][3*ps*ps],
 * K ordered (c, ky, kx) like the flattened conv weight; followed by dig_gemm_nt with bias (+pos-embed as `res`).
 * flip_from > 0: img holds flip_from images and rows of image b >= flip_from come from image b - flip_from mirrored
 * left-right (the flip-TTA batch torch.cat([img, img.flip(-1)]) of model.py:470-474 without materialising it). */
typedef struct {
    const float* img;
    float* rows;
    int B, H, W, ps, flip_from;
} dig_patchify_args;
int dig_patchify(const dig_patchify_args* a, void* stream);

/* nn.LayerNorm over the last dim (vision_transformer.py:182,187,192-194), fp32 rows */
typedef struct {
    const float* x;
    const float* gamma;
    const float* beta;
    float* out;
    long long rows;
    int D;
    float eps;
} dig_layernorm_args;
int dig_layernorm(const dig_layernorm_args* a, void* stream);

/* ViT multi-head attention that also rolls the probabilities up (vision_transformer.py:87-103; mctformer.py:302-303,
 * 310,314-315): qkv [B][N][3*heads*dh] (q|k|v, head-major inside each third) -> out [B][N][heads*dh];
 * probs are averaged over heads and ADDED into acc_all [B][N][N] (always) and acc_cls [B][ncls][N] (if non-NULL):
 * the caller zeroes both once and calls this per layer, passing acc_cls only for the last n_layers layers. */
typedef struct {
    const float* qkv;
    float* out;
    float* acc_all;
    float* acc_cls;
    int B, N, heads, dh, ncls;
} dig_vit_attn_args;
int dig_vit_attn(const dig_vit_attn_args* a, void* stream);

/* bicubic (A=-0.75, align_corners=False, coordinates from the GIVEN scale factors as F.interpolate does) resize of the
 * patch position embedding grid [g][g][D] -> [ho][wo][D]  (mctformer.py:196-217) */
typedef struct {
    const float* in;
    float* out;
    int g, D, ho, wo;
    float rscale_h, rscale_w; /* (float)(1.0 / scale_factor), exactly as F.interpolate derives its coordinate scale */
} dig_bicubic_pos_args;
int dig_bicubic_pos(const dig_bicubic_pos_args* a, void* stream);

/* CAM aggregation (mctformer.py:305-321): mtatt[b][c][p] = acc_cls[b][c][C+p]; cams = mtatt*relu(head);
 * rcams[b][c][p] = sum_q acc_all[b][C+p][C+q] * cams[b][c][q].   head: NHWC fp32 [B][P][C] (conv3x3 output). */
typedef struct {
    const float* acc_all;
    const float* acc_cls;
    const float* head;
    float* mtatt;
    float* cams;
    float* rcams;
    int B, C, P;
} dig_cam_fuse_args;
int dig_cam_fuse(const dig_cam_fuse_args* a, void* stream);

/* infer_multi post-processing (model.py:468-477 + max_norm model.py:572-578):
 *   s = up(rc)[y][x]*label + up(rc_flip)[y][W-1-x]*label   (bilinear, align_corners=False)
 *   out = relu(relu(s) - min - 1e-5) / (max - min + 1e-5), min/max of relu(s) over HxW per (b, c).   NCHW fp32. */
typedef struct {
    const float* rc;
    const float* rc_flip;
    const float* label;
    float* out;
    int B, C, hp, wp, H, W;
} dig_cam_tta_norm_args;
int dig_cam_tta_norm(const dig_cam_tta_norm_args* a, void* stream);

/* CAM label map (evaluation.py:42-50): labels[b][y][x] = argmax([thr, cam[b][0] if label>1e-5 else 0, ...]) as uint8,
 * first maximum wins (np.argmax). */
typedef struct {
    const float* cam;
    const float* label;
    unsigned char* out;
    int B, C, HW;
    float threshold;
} dig_cam_argmax_args;
int dig_cam_argmax(const dig_cam_argmax_args* a, void* stream);

/* ---- LFCA consumer of the guidance maps (networks/mctformer.py:56-166, networks/resnet101.py:195-222) ----
 * out = post(pre(x_scale * (a[b][c] * x + b[b][c])) + res_scale * res) on NHWC [B][HW][C] (C % 8 == 0).
 * aff [B][C][2] fp32 or NULL (identity); res NULL or same shape as x.  pre_act / post_act: 0 none, 1 SiLU, 2 ReLU, 3 GELU(erf).
 * Covers GroupNorm-apply + ReLU (ConvGnReLU, resnet101.py:35-38), relu(GroupNorm(y) + branch) (resnet101.py:214-222), the
 * plain GroupNorm-apply of norm_p1 (mctformer.py:164) and the LFCA0/LFCA1 average (mctformer.py:286-287). */
typedef struct {
    const void* x;
    const float* aff;
    const void* res;
    void* out;
    int B, HW, C, pre_act, post_act, dtype;
    float x_scale, res_scale;
} dig_affine_act_add_args;
int dig_affine_act_add(const dig_affine_act_add_args* a, void* stream);

/* GroupNorm statistics of an NHWC tensor that is not a conv3x3 output (norm_p1 on x_p, mctformer.py:163-164):
 * stats[b][g][tile][2] = (sum, sum of squares) over pixel slab `tile` (HW split into `tiles` contiguous slabs), the layout
 * dig_gn_finalize reads.  (C / groups) % 8 == 0, groups <= 256.  Fixed summation order: deterministic. */
typedef struct {
    const void* x;
    float* stats;
    int B, HW, C, groups, tiles, dtype;
} dig_gn_stats_args;
int dig_gn_stats(const dig_gn_stats_args* a, void* stream);

/* F.interpolate(x, size=(ho, wo), mode='bilinear', align_corners=False) on NHWC [B][hi][wi][C] -> [B][ho][wo][C],
 * optionally + res (x_p = kv + interpolate(o_p), mctformer.py:161-163). */
typedef struct {
    const void* in;
    const void* res;
    void* out;
    int B, hi, wi, ho, wo, C, dtype;
} dig_bilinear_resize_args;
int dig_bilinear_resize(const dig_bilinear_resize_args* a, void* stream);

/* mIoU bookkeeping (evaluation.py:72-88): predict, gt uint8 [n] (255 in gt = rejected boundary pixel);
 * counts uint64 [3][num_cls] are ADDED to: P (predict == i), T (gt == i), TP (gt == i and predict == gt), each over gt < 255.
 * The caller zeroes counts once and calls this per image or per batch; integer atomics, exact. */
typedef struct {
    const unsigned char* predict;
    const unsigned char* gt;
    unsigned long long* counts;
    long long n;
    int num_cls;
} dig_seg_hist_args;
int dig_seg_hist(const dig_seg_hist_args* a, void* stream);

/* storage-type conversion between the bf16 convolutions / GEMMs and the fp32 LayerNorm / attention of the LFCA consumer
 * (what torch.autocast inserts around those ops): to_f32 = 1: bf16 -> fp32, 0: fp32 -> bf16 (round to nearest even). n % 8 == 0. */
typedef struct {
    const void* in;
    void* out;
    long long n;
    int to_f32;
} dig_cast_args;
int dig_cast(const dig_cast_args* a, void* stream);

/* ==== (f)-4: backward of the classification losses through the MCTformer, and the optimiser step ============================
 * model_WSSS.update: loss_trm.backward(); self.opt_trm.step()  (models/model_ECCV24_DiG.py:424-425) for the terms
 * loss_cls / loss_cls_diff (:337-340, :406-409).  The reference leaves the backward to autograd; these are the adjoints of the
 * forward operators of vision_transformer.py:54-103,177-196 and mctformer.py:227-300, composed by dig_amd/train.py.  fp32. */

/* C[z][m*ldc + n] = alpha * sum_k A[z][m*sam + k*sak] * B[z][k*sbk + n*sbn]  (+ bias[n]) (+ the old C if accumulate),
 * batch index z = zo*z_inner + zi with an (outer, inner) element stride pair per operand.  The adjoints of nn.Linear
 * (vision_transformer.py:63-69,85-101: dX = dY W, dW = dY^T X) and of the two attention products (:92-97: dV = P^T dO,
 * dP = dO V^T, dQ = dS K, dK = dS^T Q) are each one call on the tensors where they lie - a transposed operand is a stride pair. */
typedef struct {
    const float* a;
    const float* b;
    float* c;
    const float* bias;
    long long sam, sak, sbk, sbn;
    long long sa_o, sa_i, sb_o, sb_i, sc_o, sc_i;
    long long sc_k; /* ksplit > 1: K slice s writes its partial product at c + s*sc_k (no bias / accumulate); dig_colsum adds them */
    int M, N, K, ldc, batch, z_inner, accumulate, ksplit;
    float alpha;
} dig_gemm_f32_args;
int dig_gemm_f32(const dig_gemm_f32_args* a, void* stream);

/* adjoint of nn.LayerNorm (vision_transformer.py:186,189 norm1 / norm2, eps 1e-6): x, dy [rows][D] ->
 *   dx = rstd*(g*dy - mean(g*dy) - xhat*mean(g*dy*xhat)) + add   (add: NULL or [rows][D], the residual path's gradient)
 *   part [dig_layernorm_bwd_blocks(rows)][2][D]: per-row-block sums of dy*xhat (dgamma) and dy (dbeta); dig_colsum finishes them. */
typedef struct {
    const float* x;
    const float* gamma;
    const float* dy;
    const float* add;
    float* dx;
    float* part;
    long long rows;
    int D;
    float eps;
} dig_layernorm_bwd_args;
int dig_layernorm_bwd_blocks(long long rows);
int dig_layernorm_bwd(const dig_layernorm_bwd_args* a, void* stream);

/* out[n] (+)= scale * sum_m x[m*ld + n], n < N: bias gradients of nn.Linear / Conv2d, the LayerNorm partials, and the batch sum
 * of the pos_embed / cls_token gradients (mctformer.py:231-234).  ws: [dig_colsum_chunks(M)][N] floats of workspace.
 * Two fixed-order stages, no atomics. */
typedef struct {
    const float* x;
    float* ws;
    float* out;
    long long M, ld;
    int N, accumulate;
    float scale;
} dig_colsum_args;
int dig_colsum_chunks(long long M);
int dig_colsum(const dig_colsum_args* a, void* stream);

/* adjoint of nn.GELU (erf form, vision_transformer.py:64): dx = dy * (Phi(x) + x*phi(x)); x = the fc1 output; dx may alias dy */
typedef struct {
    const float* x;
    const float* dy;
    float* dx;
    long long n;
} dig_gelu_bwd_args;
int dig_gelu_bwd(const dig_gelu_bwd_args* a, void* stream);

/* adjoint of attn.softmax(dim=-1) of the scaled logits (vision_transformer.py:93-94), in place over dp [rows][n]:
 *   dp <- scale * p * (dp - sum_j p_j dp_j)   (scale = head_dim^-0.5, so the result is the gradient of q k^T itself) */
typedef struct {
    const float* p;
    float* dp;
    const float* g; /* NULL, or [B][N][n]: added to every head's dp first - the gradient of the head-averaged, layer-summed roll-up
                     * (mctformer.py:302-315) that the rcams-consistency term sends into each layer's probabilities; rows = B*heads*N */
    long long rows;
    int n, heads, N;
    float scale;
} dig_softmax_bwd_args;
int dig_softmax_bwd(const dig_softmax_bwd_args* a, void* stream);

/* dx[b][c] = scale * d/dx F.multilabel_soft_margin_loss(x, y) = scale * (sigmoid(x) - y) / (B*C)   (model.py:337-340,406-409) */
typedef struct {
    const float* x;
    const float* y;
    float* dx;
    float scale;
    int B, C;
} dig_msm_grad_args;
int dig_msm_grad(const dig_msm_grad_args* a, void* stream);

/* adjoint of the top-3 pooled patch logits (mctformer.py:300): head [B][P][C] (the head conv's output, pre-ReLU), dpooled [B][C]
 * -> dhead [B][P][Cp] (Cp >= C: channel count padded for the conv kernels; written in full, zero where nothing flows). */
typedef struct {
    const float* head;
    const float* dpooled;
    float* dhead;
    int B, P, C, Cp;
} dig_head_pool_bwd_args;
int dig_head_pool_bwd(const dig_head_pool_bwd_args* a, void* stream);

/* one torch.optim.AdamW update of one parameter tensor (opt_trm.step(), model.py:425; timm's create_optimizer('adamw')):
 *   p *= 1 - lr*wd;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p -= step_size * m / (sqrt(v)/sqrt_bc2 + eps)
 * with step_size = lr / (1 - b1^t), sqrt_bc2 = sqrt(1 - b2^t) formed by the caller for step t. */
typedef struct {
    float* p;
    const float* g;
    float* m;
    float* v;
    long long n;
    float lr, wd, beta1, beta2, eps, step_size, sqrt_bc2;
} dig_adamw_args;
int dig_adamw(const dig_adamw_args* a, void* stream);

/* out[b][i] (+)= scale * sum_h p[b][h][i], p [B][H][n]: the roll-up of a layer's attention probabilities (torch.mean over heads,
 * sum over layers: mctformer.py:302-303,310,315) that the training forward keeps for the CAM outputs */
typedef struct {
    const float* p;
    float* out;
    long long n;
    int B, H, accumulate;
    float scale;
} dig_head_sum_args;
int dig_head_sum(const dig_head_sum_args* a, void* stream);

/* db = scale * d/db sum_hw | (max_norm(a) * (a_times_label ? label : 1) - max_norm(b)) * label |, a, b, db [B][C][P] (equal map
 * sizes), max_norm = model.py:572-578 incl. the paths through the map's maximum and minimum: the adjoint of dig_maxnorm_l1
 * (loss_cam_consistency model.py:388-390 w.r.t. cams, loss_rcams_consistency :415-417 w.r.t. the img_diff pass's rcams) */
typedef struct {
    const float* a;
    const float* b;
    const float* label;
    float* db;
    int B, C, P, a_times_label;
    float scale;
} dig_maxnorm_l1_bwd_args;
int dig_maxnorm_l1_bwd(const dig_maxnorm_l1_bwd_args* a, void* stream);

/* elementwise adjoint of cams = mtatt * relu(head) (mctformer.py:306-312; mtatt NULL: cams = relu(head), :334) for dcams [B][C][P]:
 *   G[b][c][C+j] = inv_heads * dcams * relu(head)   (G NULL or [B][C+P][C+P]: the roll-up gradient, class-token rows)
 *   dhead[b][j][c] += dcams * (mtatt or 1) * [head > 0]   (head [B][P][C], dhead [B][P][Cp]) */
typedef struct {
    const float* dcams;
    const float* mtatt;
    const float* head;
    float* G;
    float* dhead;
    int B, C, P, Cp;
    float inv_heads;
} dig_cam_fuse_bwd_args;
int dig_cam_fuse_bwd(const dig_cam_fuse_bwd_args* a, void* stream);

/* adjoint of nn.GroupNorm(1, C) (LFCA.norm_p1 mctformer.py:70,164; ResBlock_ysh's norms resnet101.py:31-38) on NHWC x [B][HW][C]:
 * dx [B][HW][C]; part [B][2][C] = per-image sums of dy*xhat (dgamma) and dy (dbeta), finished by dig_colsum over the batch */
typedef struct {
    const float* x;
    const float* gamma;
    const float* dy;
    float* dx;
    float* part;
    int B, HW, C;
    float eps;
} dig_gn1_bwd_args;
int dig_gn1_bwd(const dig_gn1_bwd_args* a, void* stream);

/* adjoint of ReLU: dx = dy * [y > 0] (y: the ReLU's output or its input); dx may alias dy */
typedef struct {
    const float* y;
    const float* dy;
    float* dx;
    long long n;
} dig_relu_bwd_args;
int dig_relu_bwd(const dig_relu_bwd_args* a, void* stream);

/* dx (+)= d/d(student) of scale * the binary-split distillation term of dig_kd_bin_loss (model.py:347-361; the teacher logits
 * enter detached): student, teacher, dx [B][C] */
typedef struct {
    const float* student;
    const float* teacher;
    float* dx;
    float T, scale;
    int accumulate, B, C;
} dig_kd_bin_grad_args;
int dig_kd_bin_grad(const dig_kd_bin_grad_args* a, void* stream);

#ifdef __cplusplus
}
#endif
#endif
