/* framepack_hip.h - C ABI of libframepack_hip.so (gfx950 / MI355X).
 *
 * The reference (wongfei2009/FramePack) has NO native interface for this path: its boundary is the
 * Python nn.Module `HunyuanVideoTransformer3DModelPacked`.  This library is the layer
 * a binding sits on: one entry point per fused kernel, raw device pointers + shapes + a hipStream_t,
 * no torch types.  Each declaration cites the reference code the kernel replaces
 * (paths relative to the reference root).  The Python host mirror is framepack_amd/model.py; the ctypes
 * binding a maintainer would add to the reference is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch caching allocator), bf16 unless noted;
 *   - "row stride" is in ELEMENTS; rows are token rows; inner dimensions are contiguous;
 *   - pointers must be 16-byte aligned and strides multiples of 8 elements unless noted;
 *   - `stream` is a hipStream_t (NULL = default stream); calls are asynchronous and allocate nothing;
 *   - return value: 0 on success, otherwise a hipError_t (hipErrorInvalidValue = 1 for a rejected
 *     argument: nothing was launched).
 */
#ifndef FRAMEPACK_HIP_H
#define FRAMEPACK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Library/ABI version (bumped when a signature changes). */
int fp_abi_version(void);

/* K1  varlen flash-attention forward, non-causal, head_dim 128.
 * Replaces diffusers_helper/models/hunyuan_video_packed.py:89-112 (attn_varlen_func; parity target is the
 * SDPA branch :97).  q,k,v,o: [total_tokens, heads, 128]; segment s covers token rows
 * cu_seqlens[s] .. cu_seqlens[s+1] (int32, device; the layout :63-77 builds), or pass NULL with nseg = 1
 * for one segment over all rows (the B == 1 branch :927-932).  max_seqlen >= the longest segment.
 * `o` may alias `q`.  o needs 8-byte alignment / stride % 4 only.  K and V must each span < 4 GiB.
 * workspace (optional, may be NULL): fp_attn_workspace_bytes() bytes of device memory (a counter + two fp32 partial
 * results of a query block per CU: 65 MiB on an MI355X; a smaller one is used as far as it reaches), 16-byte aligned.
 * With it, the launch's last work items (heads x 256-row query blocks; an eighth of the CUs' worth, or the partial last
 * round) are swept as parallel KV ranges ("parts") and merged (same result to fp32 rounding), and the parts are
 * taken from a counter at the start of the workspace, so the XCDs that finish their whole sweeps first take more of
 * them (the dies of one chip differ by 7-8 % in speed); without it the call is unchanged but the launch ends with
 * its slowest die and a partly idle last round.  The call clears the counter on `stream` (a memset node: no
 * allocation); one workspace must not serve two calls that can overlap (use one per stream). */
int64_t fp_attn_workspace_bytes(void);
int fp_attn_fwd_bf16(const void* q, const void* k, const void* v, void* o,
                     int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride,
                     const int* cu_seqlens, int nseg, int max_seqlen, int total_tokens,
                     int heads, int head_dim, float softmax_scale, void* workspace, int64_t workspace_bytes,
                     void* stream);

/* K1 for callers that cannot allocate or carry a workspace (HIP-graph capture, serving loops, harnesses that forbid
 * hipMalloc).  NO counterpart in the reference.  Same arguments, kernels and result as fp_attn_fwd_bf16 called with a
 * workspace of fp_attn_static_scratch_bytes() bytes (the split follows the number of partials the workspace holds), but the split last round's fp32 partials live in a fixed device array inside the library
 * (fp_attn_static_scratch_bytes() = 256 partials = 32.5 MiB, held by every GPU the library runs on until the process
 * ends; its per-device address is looked up once per device, at the first call with that device current).  The call
 * allocates nothing.  Limit: a shape whose last round needs more than 256 partials (only possible on a device with
 * more than 256 CUs) is rejected with hipErrorInvalidValue and nothing is launched.  All calls on one GPU share the
 * array: calls that overlap on one GPU (two streams, two threads) race; stream-ordered calls do not. */
int64_t fp_attn_static_scratch_bytes(void);
int fp_attn_fwd_bf16_static(const void* q, const void* k, const void* v, void* o,
                            int64_t q_row_stride, int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride,
                            const int* cu_seqlens, int nseg, int max_seqlen, int total_tokens,
                            int heads, int head_dim, float softmax_scale, void* stream);

/* K1, fp8 variant (opt-in; the role of the reference's quantised SageAttention branches, hunyuan_video_packed.py:92-94
 * `sageattn` and :106-107 `sageattn_varlen`): the same attention with Q, K, P and V in e4m3 on the fp8 MFMA, for one
 * segment or for the varlen segments of cu_seqlens (same meaning as in fp_attn_fwd_bf16).  The call quantises q, k, v
 * (bf16, same strides as above) into `workspace` first - one scale per head for q and for k, one per (head, channel) for
 * v, taken over ALL rows; V stored transposed in the kernel's key order, every segment with its own grid of 64-key
 * tiles - and then runs the sweep.  The result differs from fp_attn_fwd_bf16's by a few per cent (3 mantissa bits).
 * workspace: device memory, 256-byte aligned, at least fp_attn_fp8_varlen_workspace_bytes(total_tokens, heads, nseg,
 * max_seqlen) bytes (fp_attn_fp8_workspace_bytes(total_tokens, heads) for the one-segment call); o may alias q.
 * PRECONDITION (not checked: cu_seqlens lives on the device): max_seqlen >= every segment's length.  A longer segment is
 * swept over its first ceil(max_seqlen / 64) key tiles only and its query rows past ceil(max_seqlen / 256) * 256 are not
 * written (fp_attn_fwd_bf16 under the same violation reads all keys but also leaves those rows unwritten). */
int64_t fp_attn_fp8_workspace_bytes(int total_tokens, int heads);
int fp_attn_fwd_fp8(const void* q, const void* k, const void* v, void* o, int64_t q_row_stride,
                    int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride, int total_tokens, int heads,
                    int head_dim, float softmax_scale, void* workspace, int64_t workspace_bytes, void* stream);
int64_t fp_attn_fp8_varlen_workspace_bytes(int total_tokens, int heads, int nseg, int max_seqlen);
int fp_attn_fwd_fp8_varlen(const void* q, const void* k, const void* v, void* o, int64_t q_row_stride,
                           int64_t k_row_stride, int64_t v_row_stride, int64_t o_row_stride, const int* cu_seqlens,
                           int nseg, int max_seqlen, int total_tokens, int heads, int head_dim, float softmax_scale,
                           void* workspace, int64_t workspace_bytes, void* stream);

/* K2  in-place RMSNorm over head_dim (=128) of q and k, then RoPE on token rows [0, rope_rows).
 * Replaces diffusers_helper/dit_common.py:31-39 (RMSNorm_forward) + hunyuan_video_packed.py:80-86
 * (apply_rotary_emb_transposed) + the cats of :180-181.  q,k: [rows, heads, 128] sharing row_stride.
 * wq,wk: [128] norm weights; wq_txt,wk_txt: optional weights for rows >= rope_rows (norm_added_q/k of the
 * double-stream blocks, :141-142), NULL = use wq/wk.  rope: fp32 [rope_rows, 256] = cos[128] | sin[128]
 * per row (the table of :397-432).  Rounding points are the reference's. */
int fp_qk_rmsnorm_rope_bf16(void* q, void* k, int64_t row_stride, int rows, int heads, int head_dim,
                            const void* wq, const void* wk, const void* wq_txt, const void* wk_txt,
                            const float* rope, int rope_rows, float eps, void* stream);

/* K2 for Ulysses sequence parallelism (no reference counterpart: the reference is single-GPU): the same arithmetic
 * with q, k, v READ-ONLY and the results - v copied unchanged - written into `pack`, the send buffer of the head
 * exchange: bf16 [chunks][world][rows][3][hc*128] with hc = heads / world / chunks; head h goes to peer h / (heads /
 * world), chunk (h % (heads / world)) / hc.  One pass instead of K2 plus a packing copy of q|k|v. */
int fp_qk_rmsnorm_rope_pack_bf16(const void* q, const void* k, const void* v, int64_t row_stride, int rows, int heads,
                                 int head_dim, const void* wq, const void* wk, const void* wq_txt, const void* wk_txt,
                                 const float* rope, int rope_rows, float eps, void* pack, int world, int chunks,
                                 void* stream);

/* K3  y = LayerNorm(x; no affine, eps) * (1 + scale) + shift,  x,y: [rows, cols], scale,shift: [cols].
 * Replaces diffusers_helper/dit_common.py:9-10 + hunyuan_video_packed.py:453, 476, 502, 642-643.
 * cols <= 4096. */
int fp_layernorm_modulate_bf16(const void* x, int64_t x_row_stride, const void* scale, const void* shift,
                               void* y, int64_t y_row_stride, int rows, int cols, float eps, void* stream);

/* K3, affine form: y = LayerNorm(x; eps) * weight + bias,  weight, bias: [cols] - the token refiner's two
 * nn.LayerNorm(elementwise_affine=True) per block (hunyuan_video_packed.py:264, 273, run through
 * dit_common.py:9-10 LayerNorm_forward: F.layer_norm in the input dtype, i.e. fp32 arithmetic and one rounding).
 * Same kernel, limits and layout as fp_layernorm_modulate_bf16. */
int fp_layernorm_affine_bf16(const void* x, int64_t x_row_stride, const void* weight, const void* bias,
                             void* y, int64_t y_row_stride, int rows, int cols, float eps, void* stream);

/* K4  x += gate * y,  x,y: [rows, cols], gate: [cols].
 * Replaces hunyuan_video_packed.py:636-637, 649-650, 570-571. */
int fp_gate_residual_bf16(void* x, int64_t x_row_stride, const void* y, int64_t y_row_stride,
                          const void* gate, int rows, int cols, void* stream);

/* K7  out[n] = sum_k W[n,k] * f(t[k]) + bias[n],  f = SiLU (apply_silu != 0) or identity.
 * The batch-1 modulation GEMV: hunyuan_video_packed.py:451, 474, 500, 245 (Linear(SiLU(temb))).
 * W: [out_dim, in_dim] row-major (nn.Linear layout), in_dim <= 4096; bias may be NULL. */
int fp_silu_gemv_bf16(const void* t, const void* W, const void* bias, void* out,
                      int in_dim, int out_dim, int apply_silu, void* stream);

/* K10 TeaCache statistic: acc[0] = sum |a - b| (difference rounded to bf16 first), acc[1] = sum |b|.
 * Replaces hunyuan_video_packed.py:957-958.  a,b: n bf16 elements (n % 8 == 0).  acc: fp_rel_l1_acc_floats() floats
 * of device memory; the first two receive the sums, the rest is scratch for the per-block partials.  Two
 * deterministic stages, no atomics: the same inputs give the same bits (the TeaCache decision compares the result
 * with a threshold). */
int fp_rel_l1_acc_floats(void);
int fp_rel_l1_bf16(const void* a, const void* b, int64_t n, float* acc, void* stream);

/* K5  C = epilogue(A . W^T + bias): the nn.Linear of every block (hunyuan_video_packed.py:119-121,
 * 133-135, 154-156, 167-169, 552, 570 and diffusers FeedForward at :646-647).
 *   A: [M, K] row stride lda;  W: [N, K] row-major (nn.Linear layout), row stride ldw;  C: [M, N] stride ldc.
 *   epilogue: 0 = bias only; 1 = bias + tanh-GELU; 2 = bias + tanh-GELU on columns >= gelu_from only;
 *             3 = C = R + gate * (A.W^T + bias)  with R: [M, N] stride ldr (may alias C), gate: [N];
 *             4 = C = R + (A.W^T + bias).
 *   A may be given as two column segments (a2 != NULL): K = k1 + k2, the second segment read from a2
 *   (same row stride) - the cat([attn, mlp]) of :569 without the copy. */
int fp_gemm_bf16(const void* a, int64_t lda, const void* a2, int k1, const void* w, int64_t ldw,
                 const void* bias, void* c, int64_t ldc, int M, int N, int K, int epilogue, int gelu_from,
                 const void* resid, int64_t ldr, const void* gate, void* stream);

/* K5 over two row segments of one activation buffer in ONE launch: rows [0, m_split) of a multiply w0 (+ bias0, gate0),
 * rows [m_split, M) multiply w1 (+ bias1, gate1); c and resid are one buffer each, w0 and w1 have one shape and row stride.
 * Replaces the pairs of nn.Linear of a double-stream block (hunyuan_video_packed.py:537-604 with :469-535: to_q/k/v and
 * add_q/k/v_proj, to_out[0] and to_add_out, ff and ff_context), whose image rows and text rows this implementation keeps
 * in one buffer: the text segment alone is 4 - 16 row tiles - a launch that fills a fraction of the chip.  Tiles never
 * straddle m_split: bit-identical to two fp_gemm_bf16 calls (which is what runs when the operands do not meet the 4-wave
 * kernel's 16-byte alignment or a segment is empty).  epilogue 0, 1 or 3 (as fp_gemm_bf16); both biases / gates or neither. */
int fp_gemm_bf16_rows2(const void* a, int64_t lda, const void* w0, const void* w1, int64_t ldw, const void* bias0,
                       const void* bias1, void* c, int64_t ldc, int M, int m_split, int N, int K, int epilogue,
                       int gelu_from, const void* resid, int64_t ldr, const void* gate0, const void* gate1, void* stream);

/* K5 with the accumulators written out in fp32: c [M, N] fp32 = a [M, K] . w [N, K]^T + bias, a and w bf16, bias fp32
 * [N] or NULL.  The output head under `high_quality_fp32_output_for_inference` (hunyuan_video_packed.py:1048-1053:
 * activations and proj_out's weights upcast to fp32, then nn.Linear; framepack/models.py:241 always sets the flag): both
 * operands there are bf16 values in fp32 containers, whose products are exact in fp32, so the bf16 MFMA with fp32
 * accumulation computes the same sum (in another order).  K % 64 == 0, N % 8 == 0, ldc % 4 == 0, c and bias 16-byte
 * aligned. */
int fp_gemm_bf16_f32out(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, void* c,
                        int64_t ldc, int M, int N, int K, void* stream);

/* K5 + K2 in one launch: the q | k | v projection of a double-stream block (hunyuan_video_packed.py:552-563,
 * epilogue 0) or the fused in-projection of a single-stream block (:570-579, epilogue 2: tanh-GELU on columns >=
 * gelu_from) with the per-head RMSNorm of q and k and the RoPE of the rows < rope_rows (:137-160, 581-585) applied in
 * the GEMM's epilogue.  Bit-identical to fp_gemm_bf16 followed by fp_qk_rmsnorm_rope_bf16 on the same buffers (same
 * arithmetic on the same bf16-rounded values), one pass over q and k less.  q / k are the output columns [qcol0, qcol0 +
 * heads*128) / [kcol0, kcol0 + heads*128) (multiples of 128, disjoint); wq, wk, wq_txt, wk_txt, rope, rope_rows, eps
 * as in fp_qk_rmsnorm_rope_bf16.  c 16-byte aligned, ldc % 8 == 0; the other arguments as fp_gemm_bf16. */
int fp_gemm_bf16_qknorm(const void* a, int64_t lda, const void* w, int64_t ldw, const void* bias, void* c,
                        int64_t ldc, int M, int N, int K, int epilogue, int gelu_from, const void* wq, const void* wk,
                        const void* wq_txt, const void* wk_txt, const float* rope, int rope_rows, int qcol0,
                        int kcol0, int heads, int head_dim, float eps, void* stream);

/* ---- K6: fp8 linears on the gfx950 fp8 MFMA ---------------------------------------------------------------
 * Replaces the patched nn.Linear.forward of utils/fp8_optimization_utils.py (fp8_linear_forward_patch: e4m3
 * weight * scale_weight, then F.linear) with a GEMM that multiplies in e4m3.  Opt-in: quantised activations move
 * a block's output by 2-4 % relative L2, outside the 1e-2 bar of the bf16 path. */

/* K3 + K6a fused (fp8 compute mode; no reference counterpart beyond the two it fuses): the e4m3 row and its scale
 * that fp_quantize_rows_fp8 would produce from the output of fp_layernorm_modulate_bf16 (ref :453, :476, :642-643),
 * bit-identical to the two calls, in one pass over x.  q: e4m3 [rows, cols] with row stride ldq bytes; q_scale: fp32
 * [rows]; cols % 8 == 0, cols <= 4096. */
int fp_layernorm_modulate_fp8(const void* x, int64_t x_row_stride, const void* scale, const void* shift, void* q,
                              int64_t ldq, void* q_scale, int rows, int cols, float eps, void* stream);

/* x1 [M, K1] (and optionally x2 [M, K2], laid after it in K) bf16 -> q [M, K1+K2] e4m3 (OCP, max 448) and
 * scale [M] fp32 with x ~= scale[m] * q.  K1, K2 % 8 == 0; ldq (bytes = elements) % 8 == 0 (% 16 to feed fp_gemm_fp8_bf16). */
int fp_quantize_rows_fp8(const void* x1, int64_t ld1, int K1, const void* x2, int64_t ld2, int K2, void* q,
                         int64_t ldq, void* scale, int M, void* stream);

/* c [M, N] bf16 = epilogue(a_scale[m] * w_scale[n] * (a . w^T) + bias): a [M, K] and w [N, K] e4m3, K % 128 == 0,
 * lda/ldw % 16 == 0, a_scale [M] and w_scale [N] fp32 (w_scale 16-byte aligned; a per-tensor scale_weight is
 * passed broadcast to N entries).  bias/resid/gate bf16; epilogue and the remaining arguments as fp_gemm_bf16. */
int fp_gemm_fp8_bf16(const void* a, int64_t lda, const void* a_scale, const void* w, int64_t ldw,
                     const void* w_scale, const void* bias, void* c, int64_t ldc, int M, int N, int K, int epilogue,
                     int gelu_from, const void* resid, int64_t ldr, const void* gate, void* stream);

/* ---- K11: causal-3D VAE decode (fp16, channels-last [T, H, W, C] volumes) -------------------------------
 * The reference's code for this path is diffusers_helper/hunyuan.py:93-103 (vae_decode); everything under its
 * `vae.decode(...)` call is diffusers 0.33.1 AutoencoderKLHunyuanVideo (third-party, absent from the reference
 * tree, restated in oracle/vae_oracle.py: parity unpinned). */

/* Dense fp16 GEMM, same contract as fp_gemm_bf16 (epilogues 0 and 4 only): 1x1x1 convolutions
 * (post_quant_conv, conv_shortcut) and the mid-block attention's to_q/k/v, QK^T, PV and to_out. */
int fp_gemm_f16(const void* a, int64_t lda, const void* a2, int k1, const void* w, int64_t ldw,
                const void* bias, void* c, int64_t ldc, int M, int N, int K, int epilogue, int gelu_from,
                const void* resid, int64_t ldr, const void* gate, void* stream);

/* HunyuanVideoCausalConv3d 3x3x3 as an implicit GEMM: replicate padding, two frames in front and none behind.
 *   x: [Ti, Hi, Wi, Cin] (Cin % 64 == 0);  w: [Cout][27][Cin] (tap = (dt*3 + dh)*3 + dw), Cout % 8 == 0;
 *   y: [To, Ho, Wo, Cout] with Ho = Hi*up_h, Wo = Wi*up_w, To = up_t == 2 ? 2*Ti - 1 : Ti: with a factor of 2 the
 *   convolution reads the nearest-neighbour upsampling of HunyuanVideoUpsampleCausal3D (first frame not doubled
 *   in time) without materialising it.  bias: [Cout] or NULL.  resid: [To, Ho, Wo, Cout] added to the result
 *   (the resnet block's skip) or NULL; may alias y. */
int fp_conv3d_causal_f16(const void* x, const void* w, const void* bias, void* y, int Ti, int Hi, int Wi,
                         int up_t, int up_h, int up_w, int Cin, int Cout, const void* resid, void* stream);

/* The same convolution with stride (dn_t, dn_h, dn_w), each 1 or 2: the encoder's HunyuanVideoDownsampleCausal3D
 * (diffusers AutoencoderKLHunyuanVideo.encode behind ref diffusers_helper/hunyuan.py:107-110 `vae_encode`).
 *   y: [To, Ho, Wo, Cout] with To = (Ti-1)/dn_t + 1, Ho = (Hi-1)/dn_h + 1, Wo = (Wi-1)/dn_w + 1; output voxel o reads
 *   input voxels dn*o - 2 .. dn*o in time and dn*o - 1 .. dn*o + 1 in space, clamped into the volume (= the causal
 *   replicate padding followed by an unpadded strided convolution). */
int fp_conv3d_causal_strided_f16(const void* x, const void* w, const void* bias, void* y, int Ti, int Hi, int Wi,
                                 int dn_t, int dn_h, int dn_w, int Cin, int Cout, void* stream);

/* GroupNorm over a channels-last volume x [rows, C] (rows = T*H*W): stats[g] = (mean, rstd) in fp32, computed
 * in two deterministic stages.  workspace: fp_groupnorm_workspace_floats(C) floats, overwritten. */
int fp_groupnorm_workspace_floats(int C);
int fp_groupnorm_stats_f16(const void* x, int64_t rows, int C, int groups, float eps, float* workspace,
                           float* stats, void* stream);
/* y = f(GroupNorm(x; stats) * gamma + beta), f = SiLU if silu != 0 (the resnet's norm+swish), else identity
 * (the attention's group_norm).  y may alias x. */
int fp_groupnorm_apply_f16(const void* x, void* y, int64_t rows, int C, int groups, const float* stats,
                           const void* gamma, const void* beta, int silu, void* stream);

/* The mid-block attention (one head of 512 channels, HunyuanVideoMidBlock3D's Attention under
 * prepare_causal_attention_mask of diffusers 0.33.1) never materialises its S x S matrix: per frame and chunk of query
 * rows the host side computes fp32 logits against the keys that frame may see, the softmax below, and P.V (fp_gemm_f16).
 *
 * c [M, N] fp32 = a [M, K] . w [N, K]^T with fp16 operands and no bias (the logits; K % 64 == 0, N % 8 == 0,
 * lda, ldw % 8 == 0, ldc % 4 == 0, c 16-byte aligned). */
int fp_gemm_f16_f32out(const void* a, int64_t lda, const void* w, int64_t ldw, void* c, int64_t ldc, int M, int N,
                       int K, void* stream);

/* softmax(scale * s) of the fp32 logits s [rows, cols] (row stride ld floats, ld % 4 == 0, cols % 8 == 0) under the
 * frame-causal mask: query i of the call is token row0 + i and keeps the keys below
 * ((row0 + i) / q_per_frame + 1) * k_per_frame (q_per_frame = k_per_frame = h*w on one GPU; a band of rows per GPU has
 * fewer queries per frame than keys).  The fp16 probabilities are written IN PLACE over the front of each row:
 * P = (f16*)s, [rows, cols] with row stride 2*ld halves; masked and padding columns are written as 0. */
int fp_softmax_framecausal_f32_f16(void* s, int rows, int cols, int64_t ld, int row0, int q_per_frame,
                                   int k_per_frame, float scale, void* stream);

/* ---- diagnostics (no counterpart in the reference) --------------------------------------------------------
 * Select the kernel generation used by fp_gemm_bf16 / fp_attn_fwd_bf16 for same-process A/B measurements
 * (tools/ab_model.py) and for the tests of the generation that is not the default.  GEMM: 0 = first generation (8 waves
 * per workgroup), 1 = second (one wave per SIMD; default).  Attention: 1 = second generation (S^T in AGPRs), 2 = third
 * (S^T - m accumulated in VGPRs; default); + 16 = one block per split part and only a partial last round split (round
 * 2's schedule; the default takes the parts from a counter).  v < 0 only queries.  Returns the previous value.  Process-wide; the
 * switches are atomics read once per launch, so a change never tears a launch in flight on another thread.
 * fp_gemm_fp8_bf16 follows the GEMM switch too.  For the GEMM, v >> 4 (when non-zero) overrides the number of row
 * tiles per group of the tile walk (tools/gemm_xp.py scans it: FP_GEMM_VARIANT = 1 + 16 * gm); + 2: fp_gemm_bf16_rows2
 * runs its two segments as two launches; + 4: the Cout = 128 convolutions keep the 256 x 128 tile (default: 512 x 128). */
int fp_debug_gemm_variant(int v);
int fp_debug_attn_variant(int v);

#ifdef __cplusplus
}
#endif
#endif /* FRAMEPACK_HIP_H */
