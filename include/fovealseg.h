/* fovealseg.h -- C ABI of libfovealseg_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary of the FovealSeg forward/backward hot path.  The reference has no FFI: every op on
 * this path is an ATen call made from Python (models/models.py:666-1094 and the modules it drives).
 * Each entry point below replaces one such call (or a fixed group of them); the citation after each
 * prototype names the reference call site it stands in for (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into memory owned by the caller (PyTorch caching allocator);
 *     the library never allocates, frees or synchronises;
 *   - activations are NHWC fp32 (B,H,W,C) unless a prototype says NCHW; conv weights are RSCK
 *     ([R][S][Cin][Cout]) -- the physical layout behind the reference's logical (Cout,Cin,R,S) shape;
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and return immediately;
 *   - return value: 0 on success, 1001 (FS_ERR_ARG) for a rejected shape/pointer, otherwise the
 *     hipError_t of the failed launch.  Nothing is launched when an argument is rejected.
 *   - ACT codes: 0 none, 1 ReLU, 2 ReLU6.
 */
#ifndef FOVEALSEG_H
#define FOVEALSEG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* fs_stream_t;

/* ---- foveation front-end ------------------------------------------------------------------ */
/* Input pipeline step in front of the path: one decoded sample, uint8 on the device, becomes slot b of the
 * float batch.  img (H,W,Ci) uint8 in PIL memory order ('RGBA': Ci=4, 'RGB': 3); X (B,Cx,HP,WP) with Cx <= Ci leading channels,
 * X = img/255 (transforms.ToTensor) zero-padded by (left,right,top,bottom) (F.pad); mask (H,W) uint8 -> Y (B,1,HP,WP) float,
 * same padding (Y / mask nullable).  Bit-identical to DynamicFocus/e_preprocess_scripts/dataset.py:127-142 + default collate,
 * at a quarter of the H2D bytes. */
int fs_ingest_sample(const unsigned char* img, const unsigned char* mask, float* X, float* Y, int b, int H, int W, int Ci, int Cx,
                     int pad_left, int pad_right, int pad_top, int pad_bottom, fs_stream_t stream);
/* x (B,3,H,W) NCHW, focus (B,2)=(row,col) -> out (B,hs,ws,5): bilinear RGB + 2x squared gaze distance.
 * models/models.py:684-705 (gen_grid_mtx_2xHxW, sqrt/square, b_imresize, 2x cat). */
int fs_gaze_lowres_fwd(const float* x, const float* focus, float* out, int B, int H, int W, int hs, int ws, fs_stream_t stream);
/* CompressNet.forward on its own: s (B,HW,C) -> out (B,HW) = w . relu(s) + bias (the logits; C <= 32), and its backward
 * (ds, dw (C), db (1) overwritten).  models/models.py:360-372 (the plugin contract net_compress((B,24,.,.)) -> (B,1,.,.), call site :713). */
int fs_compress_fwd(const float* s, const float* w, const float* bias, float* out, int B, int HW, int C, fs_stream_t stream);
/* scratch (both backward entry points): B*(C+1) floats -- the per-image partial sums of dw and db, added in image order (every
 * cross-workgroup sum of this library is two launches, partials then an ordered sum: no result depends on workgroup timing). */
int fs_compress_bwd(const float* g, const float* s, const float* w, float* ds, float* dw, float* db, int B, int HW, int C,
                    float* scratch, fs_stream_t stream);
/* s (B,HW,C) -> xs (B,HW) = softmax_HW(w . relu(s) + bias).  models/models.py:369-372,715-723. */
int fs_compress_softmax_fwd(const float* s, const float* w, const float* bias, float* xs, int B, int HW, int C, fs_stream_t stream);
/* backward: C a multiple of 4; scratch = fs_compress_softmax_bwd_scratch_floats(B, C) floats (round 5: eight workgroups per image, one
 * record of C + 1 partial sums each). */
long fs_compress_softmax_bwd_scratch_floats(int B, int C);
int fs_compress_softmax_bwd(const float* g, const float* xs, const float* s, const float* w, float* ds, float* dw, float* db,
                            int B, int HW, int C, float* scratch, fs_stream_t stream);
/* y (B,1,H,W) -> (B,1,hs,ws) adaptive area average.  models/models.py:730. */
int fs_area_pool_fwd(const float* y, float* out, int B, int H, int W, int hs, int ws, fs_stream_t stream);
/* loss = coef * mean((minmax(xs) - minmax(t))^2) with whole-batch min/max.  stats = fs_edge_loss_stats_floats(n) floats, 32-byte
 * aligned, kept for bwd: [xs_min, xs_max, t_min, t_max, n_argmin, n_argmax] followed by the per-workgroup partial records of both
 * passes (round 5: every pass over the batch runs on up to 256 workgroups; whole-batch statistics are partials + an ordered sum).
 * models/models.py:889-891,898 (coef = 0.05 * edge_loss_scale). */
long fs_edge_loss_stats_floats(long n);
int fs_edge_loss_fwd(const float* xs, const float* t, long n, float coef, float* loss, float* stats, fs_stream_t stream);
int fs_edge_loss_bwd(const float* xs, const float* t, long n, float coef, const float* gout, float* stats, float* dxs,
                     fs_stream_t stream);
/* xs (B,hs,ws) -> grid (B,hs,ws,2)=(x,y) in [-1,1]: replication pad + Gaussian-weighted centroid + clamp.
 * g1d = the 2*pad+1 separable Gaussian taps (double).  models/models.py:594-637,819-821. */
int fs_gauss_grid_fwd(const float* xs, const double* g1d, float* grid, int B, int hs, int ws, int pad, fs_stream_t stream);
/* backward: scratch = fs_gauss_grid_bwd_scratch_floats(B, hs, ws) floats (round 5: four workgroups per image, each a band of grid columns;
 * the first of two launches hands (dp, dax, day) per grid point to the second). */
long fs_gauss_grid_bwd_scratch_floats(int B, int hs, int ws);
int fs_gauss_grid_bwd(const float* xs, const double* g1d, const float* dgrid, float* dxs, int B, int hs, int ws, int pad,
                      float* scratch, fs_stream_t stream);
/* The same pair under the other two settings of TRAIN.def_saliency_pad_mode (models/models.py:819-825): pad_mode 0 = 'replication'
 * (nn.ReplicationPad2d, what the two entry points above run), 1 = 'reflect' (F.pad(mode='reflect'); FS_ERR_ARG when pad > side - 1, which
 * torch refuses too), 2 = 'zero' (F.pad(mode='constant')).  The padded map is never materialised in any mode. */
int fs_gauss_grid_fwd_mode(const float* xs, const double* g1d, float* grid, int B, int hs, int ws, int pad, int pad_mode,
                           fs_stream_t stream);
int fs_gauss_grid_bwd_mode(const float* xs, const double* g1d, const float* dgrid, float* dxs, int B, int hs, int ws, int pad,
                           int pad_mode, float* scratch, fs_stream_t stream);
/* nn.Upsample(size=(H,W), mode='bilinear') of the deformation grid, align_corners=False: grid (B,h,w,2) -> out (B,H,W,2); the
 * task network may run at a higher resolution than the saliency map (TRAIN.task_input_size != saliency_input_size).
 * The backward needs integer factors H/h, W/w.  models/models.py:621-631. */
int fs_grid_upsample_fwd(const float* grid, float* out, int B, int h, int w, int H, int W, fs_stream_t stream);
int fs_grid_upsample_bwd(const float* g, float* dgrid, int B, int h, int w, int H, int W, fs_stream_t stream);
/* F.grid_sample(x, grid): bilinear / zeros / align_corners=False, bit-exact with ATen-CPU.
 * x (B,C,H,W) NCHW; out (B,h,w,C) if nhwc_out else (B,C,h,w).  models/models.py:909. */
int fs_grid_sample_fwd(const float* x, const float* grid, float* out, int B, int C, int H, int W, int h, int w, int nhwc_out,
                       fs_stream_t stream);
/* label = trunc(F.grid_sample(y, grid)) as int64; ysamp (nullable) gets the float sample.  models/models.py:880,951. */
int fs_grid_sample_label(const float* y, const float* grid, long long* label, float* ysamp, int B, int H, int W, int h, int w,
                         fs_stream_t stream);
/* autograd of grid_sample w.r.t. grid / w.r.t. input (scatter-add, dx overwritten). */
int fs_grid_sample_bwd_grid(const float* gout, const float* x, const float* grid, float* dgrid, int B, int C, int H, int W, int h,
                            int w, int nhwc, fs_stream_t stream);
int fs_grid_sample_bwd_input(const float* gout, const float* grid, float* dx, int B, int C, int H, int W, int h, int w, int nhwc,
                             fs_stream_t stream);
/* Inverse (un-foveating) warp, SURVEY §8(f)-3.  fs_inverse_grid: owner[b,v,u] = index yi*w+xi of the grid point that claims
 * full-resolution pixel (v,u) (-1 = hole; duplicates: the largest index wins, as ATen-CPU index_put_ does), grid_inv[b,v,u] =
 * (xi/w*2-1, yi/h*2-1), 0 in holes: feed it to fs_grid_sample_fwd(pred, grid_inv).  models/models.py:639-655,930-934.
 * fs_fill_nearest: every hole of vals (B,C,Hs,Ws) takes the value of the Euclidean-nearest claimed pixel (ties: smallest row,
 * then column) -- fillMissingValues_tensor(..., interp_mode='nearest'), models/models.py:159-286.  scratch = 2*B*Hs*Ws ints. */
int fs_inverse_grid(const float* grid, int* owner, float* grid_inv, int B, int h, int w, int Hs, int Ws, fs_stream_t stream);
int fs_fill_nearest(float* vals, const int* owner, int* scratch, int B, int C, int Hs, int Ws, fs_stream_t stream);
/* u=int((gx+1)/2*(W-1)), v=int((gy+1)/2*(H-1)) for n grid points.  models/models.py:644-645. */
int fs_inverse_index_maps(const float* grid, long long* u, long long* v, long n, int H, int W, fs_stream_t stream);

/* ---- convolution engine (implicit GEMM on the matrix cores) ------------------------------------ */
/* Arithmetic of the channel-aligned conv kernels (fp32 tensors and fp32 accumulation in every mode):
 *   0 "f32"    fp32 MFMA (v_mfma_f32_32x32x2_f32), one MFMA per product;
 *   1 "bf16x3" each fp32 operand split into three bf16 terms (exact to 24 significand bits), six bf16 MFMAs per product;
 *   2 "f16x2"  each operand scaled by a power of two and split into two fp16 terms (22-23 significand bits), three fp16
 *              MFMAs per product.
 * Default 2, or FS_CONV_PRECISION=f32|bf16x3|f16x2 in the environment at load time.
 * This is the library's ONE piece of process-global mutable state (like a BLAS math-mode switch): a host-side word read by the
 * conv entry points when they choose a kernel.  It is not a launch and is not ordered with streams; set it while no other
 * thread is inside a conv entry point (the Python layer sets it once at start-up, the tests between cases).  Everything
 * else the entry points need comes in through their arguments. */
int fs_set_conv_precision(int mode);
int fs_get_conv_precision(void);
/* Stream ordering without a host round trip: everything enqueued on `waiter` after this call runs after everything enqueued on
 * `signaller` before it (what torch's `waiter.wait_stream(signaller)` does: the reference gets the same ordering from the autograd
 * engine's stream guards).  Used to run a small layer's weight gradient beside the bwd-data chain. */
int fs_stream_wait(fs_stream_t waiter, fs_stream_t signaller);
/* Deterministic mode -- the reference asks for it with torch.backends.cudnn.deterministic = True (train_deform_semantic.py:680-681).
 * By default the bwd-weight kernels add their split-K partial tiles with fp32 atomics, whose order is the workgroups' arrival order:
 * two runs of the same step differ in the last bits of the weight gradients.  on = 1 (or FS_DETERMINISTIC=1 in the environment at
 * load time): every split writes its partial tile to its own slab of a caller-provided scratch and the slabs are summed in index
 * order; every other reduction of the library is order-fixed in both modes.  Two runs of a training step are then bit-identical.
 * Process-global host-side word like the precision mode (same rules); costs one memset + one reduce launch per bwd-weight call. */
int fs_set_deterministic(int on);
int fs_get_deterministic(void);
/* Scratch the forward (transposed=0) / bwd-data (transposed=1) entry points can use for this shape, in bytes
 * (0 = none).  In split-precision mode 3x3 / stride 1 / pad 1 convolutions with >= 32 input channels run as a
 * halo-tiled kernel that first packs the weights, already split into bf16 terms, into this scratch; without
 * scratch (ws = NULL) they fall back to the kernel that splits weights in flight.  Host-side query, no launch. */
long fs_conv2d_workspace_bytes(int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil,
                               int transposed);
/* Which kernel family fs_conv2d_fwd (transposed=0) / fs_conv2d_bwd_data (transposed=1) select for this problem under the
 * current precision mode with ws_bytes of scratch: 0 = generic 64-bit-indexed implicit GEMM (any channel count, sources of
 * 4 GB and more), 1 = plain channel-aligned implicit GEMM, 2 = halo-tiled 3x3 stride-1 kernel, 3 = tap-class kernel,
 * 4 = 1x1 / stride-1 GEMM kernel with pre-split weights (also the forward of stride >= filter layers with 64-aligned input channels,
 * as that GEMM over gathered rows), 5 = halo-tiled 3x3 stride-1 kernel with F(2,3) minimal filtering along
 * the row (even widths; 12 instead of 18 matrix steps per pixel pair), 6 = bwd-data of a 3x3 / stride-2 / pad-1 layer with the four
 * output parities in one launch, 7 = its forward with the four input parity planes in one LDS refill per chunk, 8 = the 3x3 kernel with
 * F(4,3) minimal filtering along the row (bf16x3, widths that are multiples of 4; 18 instead of 24 matrix steps per four pixels).  The
 * aligned kernels address the source with 32-bit byte offsets, so they are chosen only below 4 GB.  Host-side predicate. */
int fs_conv2d_kernel_choice(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil,
                            int transposed, long ws_bytes);
/* Weight packs that outlive the call.  The kernels of families 2-8 above start with a small launch that writes the layer's weights,
 * split into their 16-bit terms in consumption order, into ws; the weights only change at the optimiser step, so a caller may keep one
 * scratch per (layer, direction), fill it once per weight update -- on any stream, e.g. beside the first kernels of the next step
 * -- and run the convolutions on it (every nn.Conv2d on the path, e.g. models/hrnetv2_nodownsp.py:49-55).
 *   fs_conv2d_pack_persistent: 1 when ws of this problem is ONE pack that depends on (w, shape, precision mode) only, else 0.
 *   fs_conv2d_pack: run only that pack launch (FS_ERR_ARG when the predicate is 0; nothing is launched then).
 *   fs_conv2d_ws_mode(1): the calling thread's following conv entry points take ws as already packed for their problem and skip the
 *     pack launch, until fs_conv2d_ws_mode(0); returns the previous mode.  A problem whose kernel has no persistent pack fails with
 *     FS_ERR_ARG in mode 1 instead of running on stale scratch.  Host-side, thread-local. */
int fs_conv2d_pack_persistent(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil,
                              int transposed, long ws_bytes);
int fs_conv2d_pack(const float* w, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil,
                   int transposed, void* ws, long ws_bytes, const unsigned* w_amax, fs_stream_t stream);
int fs_conv2d_ws_mode(int mode);
/* Number of [Cout][2] partial-sum slabs fs_conv2d_fwd_stats writes for this shape given ws_bytes of scratch. */
int fs_conv2d_stats_slabs(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil,
                          long ws_bytes);
/* max|w| as float bits for each of nparams weight tensors stored in one arena (tensor p = sizes[p] floats at arena + offsets[p]),
 * one launch.  A caller that keeps this up to date (train.FlatAdam does, after every step) hands &out[p] to the conv entry points
 * as w_amax and saves them the per-call reduction over the weights that the f16x2 mode needs for its weight scale. */
int fs_weight_amax_segments(const float* arena, const long* offsets, const long* sizes, int nparams, unsigned* out, fs_stream_t stream);
/* F.conv2d(x, w, bias, stride, pad, dilation=dil) [+ Dropout(drop_p) keyed by drop_key when drop_p > 0].
 * ws / ws_bytes: caller-owned scratch (see fs_conv2d_workspace_bytes), may be NULL / 0.
 * w_amax: device pointer to max|w| (float bits) of this weight tensor, or NULL (then it is computed per call when needed).
 * models/hrnetv2_nodownsp.py:49-50,54-55 and every nn.Conv2d on the path. */
int fs_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin, int Ho, int Wo,
                  int Cout, int R, int S, int stride, int pad, int dil, float drop_p, uint32_t drop_key, void* ws, long ws_bytes,
                  const unsigned* w_amax, fs_stream_t stream);
/* The forward of a linear layer that ENDS a residual branch of a transformer block (transformers' modeling_segformer.py SegformerSelfOutput /
 * SegformerMixFFN.dense2 followed by `hidden_states = drop_path(...) + hidden_states`, the network /root/reference/models/segformer.py:88-100
 * runs), with the residual add in the GEMM epilogue (round 5):   y = res + DropPath_b(Dropout(conv(x, w) + bias)).
 * Dropout: the element hash of fs_conv2d_fwd (drop_p 0 = none); DropPath: the per-sample hash of fs_residual_droppath over samples of
 * rows_per_sample consecutive output pixels (droppath_p 0 = none).  Only where the 1x1 GEMM kernel runs the layer:
 * fs_conv2d_fwd_residual_ok (host only) says so, otherwise call fs_conv2d_fwd + fs_residual_droppath. */
int fs_conv2d_fwd_residual_ok(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil,
                              long rows_per_sample, long ws_bytes);
int fs_conv2d_fwd_residual(const float* x, const float* w, const float* bias, const float* res, float* y, int B, int H, int W, int Cin,
                           int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil, float drop_p, uint32_t drop_key,
                           float droppath_p, uint32_t droppath_key, long rows_per_sample, void* ws, long ws_bytes, const unsigned* w_amax,
                           fs_stream_t stream);
/* Same forward conv, additionally writing per-workgroup BatchNorm partial sums of the stored output into
 * stats = [fs_conv2d_stats_slabs(...)][Cout][2] floats (needs Cin%4==0 && Cout%4==0); finalise with
 * fs_bn_finalize_slab.  Fuses the statistics pass of F.batch_norm(training=True)
 * (lib/nn/modules/batchnorm.py:58-61) into the conv. */
int fs_conv2d_fwd_stats(const float* x, const float* w, const float* bias, float* y, float* stats, int B, int H, int W, int Cin,
                        int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil, float drop_p, uint32_t drop_key,
                        void* ws, long ws_bytes, const unsigned* w_amax, fs_stream_t stream);
/* Inference forward with the BatchNorm of eval mode folded in: z = act((conv(x, w) + bias) * scale[c] + shift[c] [+ res]),
 * scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale (lib/nn/modules/batchnorm.py:56-61 with training = False;
 * models/hrnetv2_nodownsp.py:46-62 for the residual / ReLU order), act = 0 none / 1 ReLU / 2 ReLU6.  One launch, no intermediate conv
 * output.  fs_conv2d_fwd_affine_act_ok (host only) = 1 where the kernel this shape runs on has the row epilogue, else call
 * fs_conv2d_fwd + fs_bn_eval_prepare + fs_bn_act_fwd. */
int fs_conv2d_fwd_affine_act_ok(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil,
                                long ws_bytes);
int fs_conv2d_fwd_affine_act(const float* x, const float* w, const float* bias, const float* scale, const float* shift, const float* res,
                             float* z, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil,
                             int act, void* ws, long ws_bytes, const unsigned* w_amax, fs_stream_t stream);
/* convolution_backward: input gradient / weight gradient (dw overwritten). */
int fs_conv2d_bwd_data(const float* dy, const float* w, float* dx, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int R,
                       int S, int stride, int pad, int dil, void* ws, long ws_bytes, const unsigned* w_amax, fs_stream_t stream);
/* The same call where x is the output z of a conv + BatchNorm + activation layer and this convolution is its only consumer
 * (models/hrnetv2_nodownsp.py:49-56: conv1 -> bn1 -> relu -> conv2): dx is that layer's dz, and the epilogue that holds it in registers
 * also writes the layer's BatchNorm-backward partial sums (fs_bn_bwd_partial's slab) -- bn_y / bn_mask / bn_mean / bn_invstd are the
 * layer's forward record, bn_mask NULL = no activation.  add_src (nullable): a second gradient of x to be added in the same epilogue
 * -- the residual branch of a BasicBlock (models/hrnetv2_nodownsp.py:59-62): add_src = the block's last BatchNorm's dz, add_mask its
 * activation bits (1 byte per 4 channels, NULL = unmasked) -- so autograd's add at the block input disappears as well (bn_y may be NULL
 * when only the addend is wanted).
 * fs_conv2d_bwd_data_bnsum_slabs (host only) gives the slab's row count, or 0 when the kernel this shape runs on cannot form the sums
 * (then call fs_conv2d_bwd_data + fs_bn_bwd_partial).  Kernels that can: the 3x3 stride-1 row-transform family (round 3), and since
 * round 5 the 1x1 GEMM kernel (hrnetv2_nodownsp.py:96-103: conv2 -> bn2 -> relu -> conv3 of a Bottleneck) and the one-launch 3x3 /
 * stride-2 kernel (hrnetv2_nodownsp.py:160-176: the first convolution of a two-step fuse down-path). */
int fs_conv2d_bwd_data_bnsum_slabs(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil,
                                   long ws_bytes);
int fs_conv2d_bwd_data_bnsum(const float* dy, const float* w, float* dx, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int R,
                             int S, int stride, int pad, int dil, void* ws, long ws_bytes, const unsigned* w_amax, const float* bn_y,
                             const unsigned char* bn_mask, const float* bn_mean, const float* bn_invstd, float* slab, const float* add_src,
                             const unsigned char* add_mask, fs_stream_t stream);
/* nn.Linear backward w.r.t. its parameters in one launch (the ATen addmm / sum backward behind transformers' modeling_segformer linears,
 * call sites models/segformer.py:9-11,33-37): dw[Cin][Cout] = x^T dy over `rows` rows, dbias[Cout] = column sums of dy.  Exists in the
 * bf16x3 mode for Cin, Cout multiples of 4 and >= 16 (fs_linear_bwd_weight_bias_ok == 1); otherwise FS_ERR_ARG -- callers then use
 * fs_conv2d_bwd_weight + fs_colsum.  accumulate_w / accumulate_b as `accumulate` below. */
int fs_linear_bwd_weight_bias_ok(long rows, int Cin, int Cout);
long fs_linear_bwd_weight_bias_ws_bytes(int Cin, int Cout);
int fs_linear_bwd_weight_bias(const float* x, const float* dy, float* dw, float* dbias, long rows, int Cin, int Cout, int accumulate_w,
                              int accumulate_b, void* ws, long ws_bytes, fs_stream_t stream);
/* accumulate = 0: dw is overwritten; 1: the gradient is ADDED to dw (torch's .grad accumulation; saves the memset when the caller
 * keeps a zeroed gradient arena).  ws / ws_bytes: scratch for the per-split partial tiles of deterministic mode --
 * fs_conv2d_bwd_weight_ws_bytes (0 outside that mode; ws may then be NULL).  In deterministic mode a missing or short scratch is
 * FS_ERR_ARG: the call never falls back to the atomics silently. */
long fs_conv2d_bwd_weight_ws_bytes(int Cin, int Cout, int R, int S, int stride, int pad, int dil);
int fs_conv2d_bwd_weight(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Ho, int Wo, int Cout,
                         int R, int S, int stride, int pad, int dil, int accumulate, void* ws, long ws_bytes, fs_stream_t stream);

/* ---- BatchNorm / activation / residual ------------------------------------------------------ */
/* F.batch_norm(training=True) statistics over M rows; running stats updated in place (nullable).
 * lib/nn/modules/batchnorm.py:56-61.  sums = fs_bn_stats_scratch_doubles(M, C) doubles of scratch (per-row-block records, added in
 * block order). */
long fs_bn_stats_scratch_doubles(long M, int C);
int fs_bn_stats(const float* y, long M, int C, float momentum, float eps, float* running_mean, float* running_var, float* mean,
                float* invstd, double* sums, fs_stream_t stream);
int fs_bn_finalize_slab(const float* slab, int nwg, long M, int C, float momentum, float eps, float* running_mean,
                        float* running_var, float* mean, float* invstd, fs_stream_t stream);
int fs_bn_eval_prepare(const float* running_mean, const float* running_var, int C, float eps, float* mean, float* invstd,
                       fs_stream_t stream);
/* the same statistics as the two per-channel coefficients fs_conv2d_fwd_affine_act takes: scale = gamma / sqrt(running_var + eps),
 * shift = beta - running_mean * scale */
int fs_bn_eval_affine(const float* running_mean, const float* running_var, const float* gamma, const float* beta, int C, float eps,
                      float* scale, float* shift, fs_stream_t stream);
/* out = act((y-mean)*invstd*gamma + beta [+ res]).  mask (nullable, M*C/4 bytes): bit j of byte e/4 = act'(out[e+j]) != 0, so the
 * backward passes read one byte instead of four floats of `out`.  models/hrnetv2_nodownsp.py:51-52,56-62. */
int fs_bn_act_fwd(const float* y, const float* mean, const float* invstd, const float* gamma, const float* beta, const float* res,
                  float* out, unsigned char* mask, long M, int C, int act, fs_stream_t stream);
/* Backward of the above (F.batch_norm backward + activation + dropout mask of the conv output; models/hrnetv2_nodownsp.py:46-62,
 * lib/nn/modules/batchnorm.py:56-61) in three stages; the activation derivative comes from `mask` when given, else from z (= out):
 *   fs_bn_bwd_partial : column sums sum(g), sum(g * xhat) of g = dz * act'(z) per block of rows -> slab[fs_bn_bwd_slabs(M, C)][C][2].
 *                       Skipped when the kernel that produced dz wrote the slab itself (fs_add_n_bnsum).
 *   fs_bn_bwd_finalize: slab -> dgamma, dbeta (accumulate_affine != 0: ADDED to gradient-arena targets, else overwritten) and the
 *                       per-channel coefficients coef[4][C] of the apply pass (training = 0: running statistics, no mean terms).
 *   fs_bn_bwd_apply   : dy = gradient w.r.t. the (dropped-out) conv output, dres (nullable) = gradient w.r.t. the residual input. */
int fs_bn_bwd_slabs(long M, int C);      /* host only */
int fs_bn_bwd_partial(const float* dz, const float* z, const unsigned char* mask, const float* y, const float* mean, const float* invstd,
                      long M, int C, int act, float* slab, fs_stream_t stream);
int fs_bn_bwd_finalize(const float* slab, int nslab, const float* gamma, const float* mean, const float* invstd, long M, int C,
                       int training, float* coef, float* dgamma, float* dbeta, int accumulate_affine, fs_stream_t stream);
int fs_bn_bwd_apply(const float* dz, const float* z, const unsigned char* mask, const float* y, const float* coef, long M, int C, int act,
                    float drop_p, uint32_t drop_key, float* dy, float* dres, fs_stream_t stream);

/* ---- HRNet multi-resolution fuse / concat ------------------------------------------------------ */
/* out = [relu](sum_t up(terms[t])), terms at (th[t],tw[t]) bilinearly up-sampled (align_corners=False).
 * models/hrnetv2_nodownsp.py:235-251.  terms/th/tw are HOST arrays of length nterms (<= 4). */
int fs_hr_fuse_fwd(const float* const* terms, const int* th, const int* tw, int nterms, float* out, int B, int Ho, int Wo, int C,
                   int relu, fs_stream_t stream);
int fs_relu_bwd(const float* dout, const float* out, float* g, long n, fs_stream_t stream);
/* out = a + b [+ c [+ d]] over n floats (n % 4 == 0; c, d may be NULL; out may alias a): the sum of the gradients that reach a tensor
 * with several consumers -- the block input of models/hrnetv2_nodownsp.py:46-64 (conv path + residual), the branch outputs every
 * fuse row reads (:228-252) -- which the autograd engine would otherwise form with ATen's binary add, one launch per extra consumer. */
int fs_add_n(const float* a, const float* b, const float* c, const float* d, float* out, long n, fs_stream_t stream);
/* The same sum where `out` is the gradient of a conv + BatchNorm + activation layer's output (autograd's add at the BasicBlock
 * output, models/hrnetv2_nodownsp.py:59-62): also writes that layer's BatchNorm-backward partial sums (fs_bn_bwd_partial's slab),
 * so the layer's own reduction pass is skipped.  mask / y / mean / invstd: the layer's forward record; M rows of C channels. */
int fs_add_n_bnsum(const float* a, const float* b, const float* c, const float* d, float* out, const unsigned char* mask, const float* y,
                   const float* mean, const float* invstd, long M, int C, int act, float* slab, fs_stream_t stream);
/* dst[..., coff:coff+C] = up(src); models/hrnetv2_nodownsp.py:434-442 (interpolate + cat). */
int fs_upsample_slice_fwd(const float* src, int B, int th, int tw, int C, float* dst, int Ho, int Wo, int Cdst, int coff,
                          fs_stream_t stream);
int fs_upsample_slice_bwd(const float* g, int B, int Ho, int Wo, int Cg, int coff, float* dsrc, int th, int tw, int C,
                          fs_stream_t stream);
/* Round 5: the two producers of an HRNet fuse row's gradients also form the BatchNorm-backward column sums of the layers that receive them
 * (models/hrnetv2_nodownsp.py:179-252: the last ConvBn of every fuse path has no activation, so its output gradient IS the fuse gradient):
 *   fs_relu_bwd_bnsum: g = dout * (out > 0) over (M rows, C channels, C <= 1024) and, for nterm <= 3 layers k with conv output y[k] and batch
 *     statistics mean[k] / invstd[k], slab[k][fs_bn_bwd_slabs(M, C)][C][2] = per-row-block (sum g, sum g * xhat_k) -- the input of fs_bn_bwd_finalize.
 *     y / mean / invstd / slab are HOST arrays of nterm device pointers.
 *   fs_upsample_slice_bwd_bnsum: fs_upsample_slice_bwd (even up-sampling factors) plus slab[fs_bn_bwd_slabs(B*th*tw, C)][C][2] of the layer
 *     whose output gradient dsrc is. */
int fs_relu_bwd_bnsum(const float* dout, const float* out, float* g, long M, int C, int nterm, const float* const* y, const float* const* mean,
                      const float* const* invstd, float* const* slab, fs_stream_t stream);
int fs_upsample_slice_bwd_bnsum(const float* g, int B, int Ho, int Wo, int Cg, int coff, float* dsrc, int th, int tw, int C, const float* y,
                                const float* mean, const float* invstd, float* slab, fs_stream_t stream);
/* column sums of M rows of C floats (bias gradients); accumulate != 0: ADDED to out (a gradient-arena target), else overwritten.
 * scratch = fs_colsum_scratch_floats(M, C) floats (per-row-block partial sums, added in block order). */
long fs_colsum_scratch_floats(long M, int C);
int fs_colsum(const float* x, long M, int C, float* out, int accumulate, float* scratch, fs_stream_t stream);
/* nn.MaxPool2d(k, stride, pad) on NHWC; arg = flat input pixel index of the maximum (int32), used by the backward.
 * torchvision ResNet stem behind models/deeplab.py:15. */
int fs_maxpool_fwd(const float* x, float* out, int* arg, int B, int H, int W, int C, int Ho, int Wo, int k, int stride, int pad,
                   fs_stream_t stream);
int fs_maxpool_bwd(const float* dout, const int* arg, float* dx, int B, int H, int W, int C, int Ho, int Wo, int k, int stride,
                   int pad, fs_stream_t stream);
/* nn.Dropout(p) as a stand-alone pass (same call for forward and backward); hash mask keyed by drop_key. */
int fs_dropout(const float* x, float* out, long n, float drop_p, uint32_t drop_key, fs_stream_t stream);
/* AvgPool2d((10,10)) on a 10x10 map.  models/model_utils.py:254,272. */
int fs_avgpool_fwd(const float* x, int B, int HW, int C, float* out, fs_stream_t stream);
int fs_avgpool_bwd(const float* dout, int B, int HW, int C, float* dx, fs_stream_t stream);

/* ---- C1 head tail + losses --------------------------------------------------------------------- */
/* m = sigmoid(w . x + bias) - 0.5 per pixel.  models/model_utils.py:293-298. */
int fs_mask_head_fwd(const float* x, const float* w, const float* bias, float* m, long npix, int C, fs_stream_t stream);
/* scratch = fs_mask_head_bwd_scratch_floats(npix, C) floats (per-workgroup partial sums of dw and db, added in workgroup order) */
long fs_mask_head_bwd_scratch_floats(long npix, int C);
int fs_mask_head_bwd(const float* dm, const float* m, const float* x, const float* w, float* dx, float* dw, float* db, long npix,
                     int C, float* scratch, fs_stream_t stream);
/* pred (B,K,HW) NCHW: pred[:, :K-1] = cls, pred[:, K-1] = cls[K-1]*m.  models/model_utils.py:300-306. */
int fs_pred_assemble_fwd(const float* cls, const float* m, float* pred, int B, int K, int HW, fs_stream_t stream);
int fs_pred_assemble_bwd(const float* dpred, const float* cls, const float* m, float* dcls, float* dm, int B, int K, int HW,
                         fs_stream_t stream);
/* out[7] = {dice+focal, focal, dice, acc, acc_bin_fg, acc_cls_fbg, acc_bin_fbg}; models/models.py:87-120,378-474,1057-1078.
 * accum = B * ceil(HW/1024) * (3K+7) doubles of scratch (one record per workgroup, no initialisation needed),
 * coef = 2K floats kept for the backward. */
int fs_seg_loss_fwd(const float* pred, const long long* gt, int B, int K, int HW, float gamma, float eps, double* accum,
                    float* out, float* coef, fs_stream_t stream);
int fs_seg_loss_bwd(const float* pred, const long long* gt, const float* coef, const float* gout, float* dpred, int B, int K,
                    int HW, float gamma, fs_stream_t stream);

/* ---- SegFormer (Mix-Transformer) encoder pieces; tokens are NHWC rows (B, N=H*W, C) ------------------------
 * third-party transformers==4.46.2 modeling_segformer behind models/segformer.py:9-60,87-105 */
/* nn.LayerNorm(C, eps) over the last dim of M rows; mean/rstd (M floats each) kept for the backward. */
int fs_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, long M, int C,
                     float eps, fs_stream_t stream);
/* accumulate != 0: dgamma / dbeta are ADDED to (gradient-arena targets), else overwritten.  scratch =
 * fs_layernorm_bwd_scratch_floats(M, C) floats (per-workgroup records of the two column sums, added in workgroup order). */
long fs_layernorm_bwd_scratch_floats(long M, int C);
int fs_layernorm_bwd(const float* g, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                     float* dgamma, float* dbeta, long M, int C, int accumulate, float* scratch, fs_stream_t stream);
/* The same pass with a second gradient of x added to dx (round 5): x of a pre-norm block feeds the LayerNorm and the residual add
 * (transformers' modeling_segformer.py SegformerLayer.forward, used by /root/reference/models/segformer.py:88-100), so the residual's gradient
 * arrives beside the LayerNorm's; adding it here replaces the autograd engine's add pass (104 per configs[3] step).  addend NULL = fs_layernorm_bwd. */
int fs_layernorm_bwd_add(const float* g, const float* x, const float* gamma, const float* mean, const float* rstd, const float* addend,
                         float* dx, float* dgamma, float* dbeta, long M, int C, int accumulate, float* scratch, fs_stream_t stream);
/* exact (erf) GELU. */
int fs_gelu_fwd(const float* x, float* y, long n, fs_stream_t stream);
int fs_gelu_bwd(const float* g, const float* x, float* dx, long n, fs_stream_t stream);
/* nn.GELU followed by nn.Dropout(p) (SegformerMixFFN: intermediate_act_fn, dropout; transformers 4.46.2, models/segformer.py:2,88-100) in
 * one pass, and the backward of the pair in one pass: the mask of fs_dropout with the same key on the same element index.  n < 2^32. */
int fs_gelu_dropout_fwd(const float* x, float* y, long n, float drop_p, uint32_t key, fs_stream_t stream);
int fs_gelu_dropout_bwd(const float* g, const float* x, float* dx, long n, float drop_p, uint32_t key, fs_stream_t stream);
/* depthwise Conv2d(C, C, 3, 1, 1, groups=C) on NHWC, weight (C,1,3,3); flip=1 gives the input gradient. */
int fs_dwconv3_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int C, int flip,
                   fs_stream_t stream);
/* weight gradient: ws = fs_dwconv3_wgrad_lanes(B, H, W, C) * 9 * C floats of scratch (per-thread partial sums, added up by a second
 * launch: no atomics, no memset); accumulate != 0: dw is ADDED to, else overwritten. */
int fs_dwconv3_wgrad_lanes(int B, int H, int W, int C);      /* host only */
int fs_dwconv3_bwd_weight(const float* x, const float* dy, float* dw, float* ws, int B, int H, int W, int C, int accumulate,
                          fs_stream_t stream);
/* ... with the bias gradient db[C] = column sums of dy formed by the same two launches (the kernel reads every dy element exactly once
 * already; a separate column-sum pass re-read 131 MB per Mix-FFN block of configs[3]): ws = lanes * 10 * C floats. */
int fs_dwconv3_bwd_weight_bias(const float* x, const float* dy, float* dw, float* db, float* ws, int B, int H, int W, int C,
                               int accumulate_w, int accumulate_b, fs_stream_t stream);
/* out = x + DropPath_p(y) (per-sample keep, hash keyed); x NULL -> out = scaled y (the backward of the y branch). */
int fs_residual_droppath(const float* x, const float* y, float* out, long n, long per_sample, float drop_p, uint32_t key,
                         fs_stream_t stream);
/* ... and that form's gradient with respect to the layer's output in one pass: dz = Dropout-mask(DropPath-scale_b * g), the two masks in
 * the order fs_residual_droppath (x = NULL) followed by fs_dropout apply them.  n < 2^32, n and per_sample multiples of 4. */
int fs_droppath_dropout_bwd(const float* g, float* dz, long n, long per_sample, float droppath_p, uint32_t droppath_key, float drop_p,
                            uint32_t drop_key, fs_stream_t stream);
/* A convolution whose filter has more taps than the aligned kernels take (SegFormer's 7x7 patch embedding on 3 channels and 8x8 stride-8
 * sequence-reduction conv: transformers 4.46.2 SegformerOverlapPatchEmbeddings / SegformerEfficientSelfAttention.sr, models/segformer.py:
 * 9-11,33-37) as patch rows: col[B*Ho*Wo][Kp], element (r, s, c) of the k x k patch in the RSCK weight's row order, zeros outside the image
 * and in the padding columns (Kp % 4 == 0) -- the layer is then one linear layer.  fs_fold is the adjoint (dx overwritten). */
int fs_unfold(const float* x, float* col, int B, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo, int Kp, fs_stream_t stream);
int fs_fold(const float* col, float* dx, int B, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo, int Kp, fs_stream_t stream);
/* softmax(q k^T * scale) (dropout p) v per head on the matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32
 * accumulate); head_dim 64, any number Nk of sequence-reduced key/value tokens (streamed in chunks of 64, online softmax);
 * q/o (B,N,heads*64), k/v (B,Nk,heads*64); lse = B*heads*N floats (log-sum-exp per query row, kept for the backward).
 * Replaces SegformerEfficientSelfAttention's matmul-softmax-dropout-matmul (transformers 4.46.2, models/segformer.py:2,88-100). */
int fs_attention_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int B, int N, int Nk, int heads,
                     float scale, float drop_p, uint32_t key, fs_stream_t stream);
/* The same forward in split precision (bf16x3: q*scale, k, v and the probabilities as three bf16 planes = 24-bit operands, six
 * v_mfma_f32_32x32x16_bf16 per product, fp32 accumulation and softmax) -- the arithmetic of the conv engine's headline mode, 3/8 of the
 * matrix-pipe time of the exact kernel.  ws = fs_attention_split_ws_bytes(B, Nk, heads) bytes of scratch (K / V^T planes, written by a
 * pre-pass inside the call).  Same dropout hash and element index as fs_attention_fwd.  csrc/attention_split.hip. */
long fs_attention_split_ws_bytes(int B, int Nk, int heads);
/* mask (nullable) = fs_attention_mask_words(B, N, Nk, heads) words: with drop_p > 0 the forward leaves the keep decisions there, one bit per
 * (query, key) -- row (b*heads + head)*N + q, word key >> 5, bit key & 31 -- and the split backward reads them instead of hashing again. */
long fs_attention_mask_words(int B, int N, int Nk, int heads);
int fs_attention_fwd_split(const float* q, const float* k, const float* v, float* o, float* lse, unsigned* mask, void* ws, long ws_bytes,
                           int B, int N, int Nk, int heads, float scale, float drop_p, uint32_t key, fs_stream_t stream);
/* Backward of the above: dq, dk, dv overwritten.  o = the forward's output, go = its gradient, scratch = B*heads*N floats. */
int fs_attention_bwd(const float* q, const float* k, const float* v, const float* o, const float* go, const float* lse, float* dq,
                     float* dk, float* dv, float* scratch, int B, int N, int Nk, int heads, float scale, float drop_p, uint32_t key,
                     fs_stream_t stream);
/* The backward in split precision (bf16x3): dQ with the query on the MFMA lane (transposed score tile, dS fed to the K^T product from
 * registers), dK and dV with the key on the lane (P~ / dS fed to the dO^T / Q^T products from registers, the query-slice images read by rows
 * and transposed), no LDS round trip for P or dS.  ws = fs_attention_bwd_split_ws_bytes(B, Nk, heads) bytes of scratch (the dQ kernel's
 * K / V / K^T planes); mask = the forward's keep words or NULL (the kernels then hash again).  fs_attention_bwd_dq_split / _dkv_split are
 * the two parts alone (D = rowsum(dO * O), B*heads*N floats). */
long fs_attention_bwd_split_ws_bytes(int B, int Nk, int heads);
int fs_attention_bwd_dq_split(const float* q, const float* k, const float* v, const float* go, const float* lse, const float* D,
                              const unsigned* mask, float* dq, void* ws, long ws_bytes, int B, int N, int Nk, int heads, float scale,
                              float drop_p, uint32_t key, fs_stream_t stream);
/* parts (nullable): 2 * 8 * B * Nk * heads * 64 floats for the partial dK / dV of up to 8 splits of the query range (summed by a small
 * kernel; without it the range is not split).  fs_attention_bwd_split carves it from its ws at fs_attention_bwd_split_parts_offset. */
long fs_attention_bwd_split_parts_offset(int B, int Nk, int heads);
int fs_attention_bwd_dkv_split(const float* q, const float* k, const float* v, const float* go, const float* lse, const float* D,
                               const unsigned* mask, float* dk, float* dv, float* parts, int B, int N, int Nk, int heads, float scale,
                               float drop_p, uint32_t key, fs_stream_t stream);
int fs_attention_bwd_split(const float* q, const float* k, const float* v, const float* o, const float* go, const float* lse,
                           const unsigned* mask, float* dq, float* dk, float* dv, float* scratch, void* ws, long ws_bytes, int B, int N,
                           int Nk, int heads, float scale, float drop_p, uint32_t key, fs_stream_t stream);

/* ---- optimiser ------------------------------------------------------------------------------- */
/* torch.optim.Adam(weight_decay) step over a flat fp32 arena of n (multiple of 4) elements; step >= 1;
 * grad_scale multiplies the gradient first (1/world_size).  train_deform_semantic.py:115-123,271-288. */
int fs_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int step, float grad_scale, fs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
