/*
 * w2l_hip.h — C ABI of libw2l_hip.so: the MI355X (gfx950) kernels behind the Wav2Lip hot path.
 *
 * Every entry point is `extern "C"`, takes plain device pointers + sizes and a HIP stream
 * (passed as `void*`, i.e. a `hipStream_t`; NULL = the null stream), enqueues work on that
 * stream and returns without synchronising.  Return value: 0 = ok, negative = error; the
 * message of the last error on the calling thread is returned by w2l_last_error().
 * Nothing is thrown across the ABI.  Caller owns every tensor; the library owns only the
 * handles it creates (packed weights, tap tables, plans) and frees them in *_destroy.
 *
 * Activation layout is NHWC fp32 with an explicit per-pixel channel stride ("cs", in floats):
 * a tensor argument is (pointer to channel 0 of pixel 0, cs).  That is how the skip-concats
 * of the generator (reference models/wav2lip.py:104-114) disappear: producers write straight
 * into channel slices of the concat buffer.
 *
 * Reference interfaces replaced (paths relative to the Rudrabha/Wav2Lip checkout):
 *   w2l_conv_*            models/conv.py:5-19 (Conv2d = conv+BN(+res)+ReLU), :21-31 (nonorm_Conv2d),
 *                         :33-44 (Conv2dTranspose); bare conv + Sigmoid heads models/wav2lip.py:84-85,152
 *   w2l_bn_fold           the eval-mode nn.BatchNorm2d inside those blocks (models/conv.py:10,40)
 *   w2l_nchw_to_nhwc,
 *   w2l_nhwc_to_nchw      the layout glue at inference.py:259-260,265 and models/wav2lip.py:93-94,118-120
 *   w2l_datagen_pack      inference.py:133-143 (mask lower half, concat, /255.) + :259 (transpose, f64->f32)
 *   w2l_frames_to_u8      inference.py:265,269 (x255, astype(uint8) truncation, NHWC)
 *   w2l_crop_resize_u8    inference.py:121-126 (face crop -> cv2.resize to 96x96)
 *   w2l_resize_paste_u8   inference.py:270-271 (cv2.resize of the generated crop to the box size + paste into the frame)
 *   w2l_melspectrogram    audio.py:45-51 (preemphasis, STFT, mel basis, dB, normalise)
 *   w2l_mel_gather        inference.py:231-240 (16-frame mel windows at host-computed starts)
 *   w2l_resample_sinc     audio.py:9-10 (librosa.core.load's sample-rate conversion: resampy 'kaiser_best' sinc interpolation)
 *   w2l_l2norm_rows       models/syncnet.py:62-63 (F.normalize(p=2, dim=1))
 *   w2l_cosine_bce        wav2lip_train.py:179-184 (cosine_similarity + BCELoss)
 *   w2l_plan_*            the per-batch forward loop inference.py:262-263 -> models/wav2lip.py:87-125
 *
 * Training side (the torch autograd graph behind loss.backward() / optimizer.step() at wav2lip_train.py:229-230,
 * color_syncnet_train.py:164-165, hq_wav2lip_train.py:231-232,256-257):
 *   w2l_conv_update       re-pack a layer after an optimiser step (weights change every step)
 *   w2l_conv_wgrad        weight gradient of nn.Conv2d / nn.ConvTranspose2d (models/conv.py:8,24,36)
 *                         (the data gradient is w2l_conv_forward on the transposed geometry)
 *   w2l_bn_train_stats,
 *   w2l_affine_act,
 *   w2l_bn_train_bwd      nn.BatchNorm2d in batch-statistics mode + residual + ReLU, forward and backward
 *                         (models/conv.py:10-12,17-19,40-43)
 *   w2l_act_bwd           backward of ReLU / LeakyReLU / Sigmoid (+ eval-mode BN scale) (models/conv.py:12,27;
 *                         models/wav2lip.py:85,152)
 *   w2l_col_sum           bias gradients
 *   w2l_l1_mean/_bwd      nn.L1Loss (wav2lip_train.py:191,227)
 *   w2l_cosine_bce_bwd    backward of cosine_loss (wav2lip_train.py:179-184)
 *   w2l_l2norm_bwd        backward of F.normalize (models/syncnet.py:62-63)
 *   w2l_bce_bwd           backward of F.binary_cross_entropy (models/wav2lip.py:171, hq_wav2lip_train.py:249,253)
 *   w2l_adam_*            optim.Adam (wav2lip_train.py:359, hq_wav2lip_train.py:418-421)
 *   w2l_shifted_pdist     calc_pdist of the LSE-D / LSE-C scorer (evaluation/scores_LSE/SyncNetInstance_calc_scores.py:19-31)
 */
#ifndef W2L_HIP_H
#define W2L_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define W2L_OK 0
#define W2L_ERR_ARG (-1)      /* bad argument / unsupported geometry */
#define W2L_ERR_HIP (-2)      /* a HIP runtime call failed */
#define W2L_ERR_NOMEM (-3)

/* activation applied after y = acc*scale + shift (+ residual) */
#define W2L_ACT_NONE 0
#define W2L_ACT_RELU 1        /* models/conv.py:12  */
#define W2L_ACT_SIGMOID 2     /* models/wav2lip.py:85 */
#define W2L_ACT_LEAKY 3       /* LeakyReLU(0.01), models/conv.py:27 */

/* arithmetic of a conv layer's contraction (tensors in HBM are fp32 either way) */
#define W2L_PREC_F32 0        /* exact fp32 products on the fp32 matrix cores (default; the inference parity path) */
#define W2L_PREC_BF16 1       /* operands rounded to bf16 (RNE) inside the kernel, fp32 accumulate, bf16 matrix cores:
                                 the mixed precision BASELINE configs 4/5 name for training */

const char* w2l_last_error(void);
/* library/ABI version, bumped on any signature change */
int w2l_abi_version(void);
/* number of visible HIP devices (<=0: none/err) and the gcnArchName of `dev` copied into buf */
int w2l_device_count(void);
int w2l_device_arch(int dev, char* buf, size_t buflen);

/* ---------------------------------------------------------------- fused convolution layers */

typedef struct w2l_conv_geom {
    int transposed;           /* 0: nn.Conv2d weight [cout][cin][kh][kw]; 1: nn.ConvTranspose2d weight [cin][cout][kh][kw] */
    int cin, cout;
    int kh, kw;
    int sh, sw;               /* stride */
    int ph, pw;               /* padding */
    int oph, opw;             /* output_padding (transposed only) */
    int act;                  /* W2L_ACT_* */
} w2l_conv_geom;

typedef struct w2l_conv w2l_conv_t;

/* Build a layer: packs `weight` (torch layout, device, fp32) into the kernel's K-major layout
 * (one slab per output phase for transposed convs), copies scale/shift ([cout], device).
 * y = act( conv(x, weight) * scale + shift (+ res) ).  The input buffer must provide
 * w2l_conv_cin_padded(cin) readable channels per pixel with the pad channels ZERO. */
int w2l_conv_create(const w2l_conv_geom* g, const float* weight, const float* scale,
                    const float* shift, void* stream, w2l_conv_t** out);
int w2l_conv_destroy(w2l_conv_t* c);
/* Re-pack the layer for new parameters (asynchronous on `stream`, no allocation, no synchronisation): any of
 * weight / scale / shift may be NULL = unchanged.  The caller's tensors must stay alive until the stream reaches
 * the copies (stream-ordered, as for any other kernel argument). */
int w2l_conv_update(w2l_conv_t* c, const float* weight, const float* scale, const float* shift, void* stream);
int w2l_conv_cin_padded(int cin);                       /* roundup(cin, 4) */
int w2l_conv_out_hw(const w2l_conv_geom* g, int H, int W, int* Ho, int* Wo);
/* Enqueue one fused layer.  x: [N,H,W,x_cs] fp32 (first cin_padded channels of each pixel used);
 * y: [N,Ho,Wo,y_cs] (first cout channels of each pixel written); res: optional [N,Ho,Wo,res_cs]
 * residual added before the activation (NULL = none).  res may alias x. */
int w2l_conv_forward(const w2l_conv_t* c, void* stream, int N, int H, int W,
                     const float* x, int x_cs, float* y, int y_cs,
                     const float* res, int res_cs);
/* Fuse a following 1x1 convolution + activation into the layer's epilogue (the generator's RGB head, reference
 * models/wav2lip.py:83-85: Conv2d(80,32,3)+BN+ReLU -> nn.Conv2d(32,head_c,1) -> Sigmoid): after this call w2l_conv_forward
 * writes y[pix][o] = head_act( sum_c head_weight[o][c] * act(...)[c] + head_bias[o] ), o < head_c <= 4, instead of the cout
 * channels (which never reach HBM).  head_weight [head_c][cout], head_bias [head_c] or NULL: device fp32.  Needs
 * cout % 4 == 0 and cout <= 128; residual inputs are not supported on a layer with a head. */
int w2l_conv_attach_head(w2l_conv_t* c, const float* head_weight, const float* head_bias, int head_c, int head_act,
                         void* stream);
/* nominal multiply-accumulates of one forward at (N,H,W) — the reference's direct-conv count */
long long w2l_conv_macs(const w2l_conv_geom* g, int N, int H, int W);
/* select the contraction arithmetic of this layer (W2L_PREC_*); Winograd is only used with W2L_PREC_F32 */
int w2l_conv_set_precision(w2l_conv_t* c, int precision);
/* tile configuration override for tuning/tests: -1 = automatic */
int w2l_conv_set_tile(w2l_conv_t* c, int tile_id);
int w2l_conv_num_tiles(void);

/* scale = gamma/sqrt(var+eps); shift = (bias-mean)*scale + beta.  Any of bias/gamma/beta/mean/var may
 * be NULL (treated as 0 / 1 / 0 / 0 / 1 with eps ignored when var is NULL): covers conv+BN (models/conv.py:8-11),
 * bare conv with bias (models/wav2lip.py:84) and nonorm conv (models/conv.py:24-26). */
int w2l_bn_fold(void* stream, int C, const float* bias, const float* gamma, const float* beta,
                const float* mean, const float* var, float eps, float* scale, float* shift);

/* ---------------------------------------------------------------- layout / data path */

/* x [N,C,H,W] fp32 -> y [N,H,W,y_cs] channels [0,C); channels [C,c_zero_to) of each pixel are zero-filled */
int w2l_nchw_to_nhwc(void* stream, int N, int C, int H, int W, const float* x, float* y, int y_cs,
                     int c_zero_to);
/* x [N,H,W,x_cs] (first C channels) -> y [N,C,H,W] */
int w2l_nhwc_to_nchw(void* stream, int N, int C, int H, int W, const float* x, int x_cs, float* y);

/* faces u8 [N,S,S,3] (BGR crops already SxS) -> x fp32 [N,S,S,y_cs]:
 * ch0-2 = face with rows >= S/2 zeroed, ch3-5 = face, all (float)((double)v/255.0); ch6..c_zero_to-1 = 0.
 * inference.py:136-139,259 */
int w2l_datagen_pack(void* stream, int N, int S, const uint8_t* faces, float* y, int y_cs, int c_zero_to);
/* pred fp32 [N,H,W,x_cs] (first 3 ch, values in [0,1]) -> u8 [N,H,W,3] = (uint8)(v*255.f) (truncation).
 * inference.py:265,269 */
int w2l_frames_to_u8(void* stream, int N, int H, int W, const float* x, int x_cs, uint8_t* y);

/* Box tensors below: int32 [B][4] = (y1, y2, x1, x2) per item, device memory, 16-byte aligned, 0 <= y1 < y2 <= H and
 * 0 <= x1 < x2 <= W (validated by the caller); frame_idx int32 [B] selects the frame of each item (NULL = item b uses
 * frame b).  Resizing follows cv::resize(INTER_LINEAR) for CV_8UC3 (OpenCV 4.1.0 fixed-point path, see csrc/resize.hip). */

/* out u8 [B,S,S,3] = resize(frames[frame_idx[b]][y1:y2, x1:x2], (S,S)):  inference.py:121-126 */
int w2l_crop_resize_u8(void* stream, int B, const uint8_t* frames, int H, int W, const int32_t* frame_idx,
                       const int32_t* boxes, int S, uint8_t* out);
/* dst u8 [B,Hd,Wd,3] = resize(src u8 [B,Hs,Ws,3], (Wd, Hd)): the `--resize_factor` step of inference.py:202-203 */
int w2l_resize_u8(void* stream, int B, const uint8_t* src, int Hs, int Ws, uint8_t* dst, int Hd, int Wd);
/* frames[frame_idx[b]][y1:y2, x1:x2] = resize(pred[b] (u8 [S,S,3]), (x2-x1, y2-y1)), in place:  inference.py:270-271.
 * max_box_pixels >= the largest (y2-y1)*(x2-x1) of the batch (sizes the launch). */
int w2l_resize_paste_u8(void* stream, int B, const uint8_t* pred, int S, const int32_t* boxes, const int32_t* frame_idx,
                        uint8_t* frames, int H, int W, int max_box_pixels);

/* ---------------------------------------------------------------- S3FD face detector glue (face_detection/detection/sfd/)
 * The detector's convolutions are w2l_conv_* layers (bias + ReLU, no BatchNorm); these are the ops between them. */

/* bgr u8 [npix][3] -> y fp32 [npix][y_cs]: RGB order (api.py:62 images[..., ::-1]) minus (104,117,123) (detect.py:57),
 * channel 3 zero when y_cs %% 4 == 0 */
int w2l_s3fd_pack(void* stream, long long npix, const uint8_t* bgr, float* y, int y_cs);
/* F.max_pool2d(x, 2, 2) on NHWC: y [N,H/2,W/2,y_cs] (net_s3fd.py:75-97); C %% 4 == 0 */
int w2l_maxpool2x2(void* stream, int N, int H, int W, int C, const float* x, int x_cs, float* y, int y_cs);
/* L2Norm (net_s3fd.py:6-19): y[row][c] = x[row][c] / (sqrt(sum_c x^2) + 1e-10) * weight[c] */
int w2l_l2norm_scale(void* stream, long long rows, int C, const float* x, int x_cs, const float* weight, float* y, int y_cs);
/* One detection level: cls [B,FH,FW,cls_cs] (ncls = 4: max-out background over the first three, net_s3fd.py:123-126;
 * ncls = 2), reg [B,FH,FW,reg_cs] -> out [B][FH*FW][5] = (x1, y1, x2, y2, softmax score) with the prior of `stride`
 * (detect.py:66-84, bbox.py:91-108, variances 0.1 / 0.2) */
int w2l_s3fd_decode(void* stream, int B, int FH, int FW, int stride, const float* cls, int cls_cs, int ncls, const float* reg,
                    int reg_cs, float* out);
/* Candidate gate + greedy non-maximum suppression per image (sfd_detector.py:39-45: `bboxlist[:, 4] > 0.05`, then bbox.py:44-64
 * `nms(dets, thresh)`): table [B][P][5] = (x1, y1, x2, y2, score) rows (the concatenated w2l_s3fd_decode levels, or any box
 * list); rows with score > gate compete, best score first (equal scores: the later row first - the reference leaves that order
 * to numpy's unstable argsort); a box is dropped when its overlap with a kept one is not <= thresh, the overlap being the
 * reference's float32 expression evaluated operation by operation.  keep [B][P] receives the kept ROW indices of each image in
 * selection order (score descending), counts [B] how many.  scratch: >= 12 * B * P bytes, 8-byte aligned.  P is not bounded;
 * an image with more than 262144 rows ABOVE THE GATE (the alive bitmap of the greedy pass lives in LDS) gets counts = -1 and no
 * keep list: the caller runs that image's suppression elsewhere (wav2lip_amd/face_detection/s3fd.py does, on the host).  Scores
 * of either sign order as floats do. */
int w2l_s3fd_nms(void* stream, int B, int P, const float* table, float gate, float thresh, int* keep, int* counts,
                 void* scratch, long long scratch_bytes);

/* ---------------------------------------------------------------- audio */

/* A mel context owns the device copies of the constant tables: the Slaney mel basis fp32 [80][401] and the
 * periodic Hann window f64 [800] (both built once by the host, wav2lip_amd/audio.py, exactly as
 * librosa.filters.mel / scipy.signal.get_window build them) plus the DFT twiddles.
 * hparams fixed to hparams.py:32-69 (n_fft 800, hop 200, win 800, sr 16000, 80 mels, fmin 55, fmax 7600,
 * preemphasis 0.97, ref 20 dB, min -100 dB, symmetric, max_abs 4). */
typedef struct w2l_mel w2l_mel_t;
int w2l_mel_create(const float* mel_basis_host, const double* window_host, w2l_mel_t** out);
int w2l_mel_destroy(w2l_mel_t* m);
int w2l_mel_num_frames(long long nsamples);            /* 1 + nsamples/200 */
/* wav fp32 [nsamples] (device) -> mel fp32 [80][T] row-major (as audio.melspectrogram returns it) */
int w2l_melspectrogram(const w2l_mel_t* m, void* stream, const float* wav, long long nsamples, float* mel);
/* mel [80][T] + starts int32[B] (device) -> out fp32 [B][80][16][out_cs] channel 0 (others zero up to c_zero_to) */
int w2l_mel_gather(void* stream, const float* mel, int T, const int32_t* starts, int B, float* out,
                   int out_cs, int c_zero_to);

/* Sample-rate conversion of audio.load_wav (audio.py:9-10 -> librosa.core.load(path, sr=16000) -> resampy.resample(...,
 * filter='kaiser_best')).  x fp32 [n_in], tr f64 [n_out] = the interpolator's time register at every output sample (the host
 * accumulates 1/sample_ratio by repeated float64 addition, as the reference loop does), win / delta f64 [nwin] = the filter's
 * half window (pre-scaled by sample_ratio when < 1) and its first differences, num_table = table samples per zero crossing;
 * y fp32 [n_out].  All pointers are device memory.  Bit-exact with the loop order and the per-term float32 rounding of
 * resampy's resample_f. */
int w2l_resample_sinc(void* stream, const float* x, int n_in, const double* tr, int n_out, double sample_ratio,
                      const double* win, const double* delta, int nwin, int num_table, float* y);

/* ---------------------------------------------------------------- SyncNet tail / losses */

/* x [N,x_cs] first C -> y [N,C] = x / max(||x||_2, 1e-12) */
int w2l_l2norm_rows(void* stream, int N, int C, const float* x, int x_cs, float* y);
/* a,v [N,C] -> cos[N] (eps 1e-8, F.cosine_similarity) and, if y != NULL, loss[0] = mean BCE(cos, y) */
int w2l_cosine_bce(void* stream, int N, int C, const float* a, const float* v, const float* y,
                   float* cos_out, float* loss_out);

/* p,y [N] -> loss[0] = mean( -(y*max(log p,-100) + (1-y)*max(log(1-p),-100)) )  (nn.BCELoss / F.binary_cross_entropy:
 * models/wav2lip.py:171, hq_wav2lip_train.py:249,253) */
int w2l_bce_mean(void* stream, int N, const float* p, const float* y, float* loss_out);

/* ---------------------------------------------------------------- training: conv weight gradient */

/* dweight (torch layout of `g`: [cout][cin][kh][kw], or [cin][cout][kh][kw] when transposed; device fp32, fully
 * overwritten) = d loss / d weight given the layer input x [N,H,W,x_cs] and the gradient dz [N,Ho,Wo,dz_cs] of the
 * convolution output (before BN / activation).  Both buffers must expose roundup(channels,4) readable channels per
 * pixel with ZERO pad channels, 16-byte aligned, channel strides multiples of 4.  Deterministic (fixed-order split-K). */
int w2l_conv_wgrad(const w2l_conv_geom* g, void* stream, int N, int H, int W, const float* x, int x_cs,
                   const float* dz, int dz_cs, float* dweight);

/* the same with the contraction arithmetic selectable (W2L_PREC_BF16: operands rounded to bf16 in the kernel, fp32
 * accumulate, bf16 matrix cores) */
int w2l_conv_wgrad_prec(const w2l_conv_geom* g, void* stream, int N, int H, int W, const float* x, int x_cs,
                        const float* dz, int dz_cs, float* dweight, int precision);

/* ---------------------------------------------------------------- training in bf16 (BASELINE configs[3] / [4])
 * The bf16-STORAGE training path: activations, pre-BatchNorm conv outputs and their gradients live in HBM as NHWC bf16
 * (channel strides in ELEMENTS, multiples of 8, pad channels zero; 16-byte aligned pointers); master weights, weight
 * gradients, BatchNorm statistics / parameters, losses and Adam stay fp32.  Every contraction multiplies bf16 operands on
 * v_mfma_f32_32x32x16_bf16 and accumulates in fp32; every tensor is rounded to bf16 (RNE) exactly once, on its way out. */

/* A bf16-storage conv layer (forward of models/conv.py:8,24,36 and - on the transposed geometry over the same weight tensor -
 * its data gradient): y = act( conv(x, weight) * scale + shift (+ res) ), act = g->act.  `weight` is the fp32 master tensor in
 * torch layout; create / update pack it to bf16 (asynchronous on `stream`; call update after every optimiser step). */
typedef struct w2l_convb w2l_convb_t;
int w2l_convb_create(const w2l_conv_geom* g, const float* weight, void* stream, w2l_convb_t** out);
int w2l_convb_update(w2l_convb_t* c, const float* weight, void* stream);
/* The same re-pack for n layers in ONE launch (what optimizer.step() in wav2lip_train.py:231 invalidates:
 * every layer's slabs).  weights[i] is the fp32 master tensor of handles[i]; the device-side tables are cached per (handle,
 * weight pointer) list, so steady-state training steps cost one kernel launch.  Bytes written == n calls of w2l_convb_update. */
int w2l_convb_update_many(int n, w2l_convb_t* const* handles, const float* const* weights, void* stream);
int w2l_convb_destroy(w2l_convb_t* c);
/* x bf16 [N,H,W,x_cs], y bf16 [N,Ho,Wo,y_cs] (channels [0, roundup(cout,8)) of each pixel written, pad channels zero), res
 * optional bf16 [N,Ho,Wo,res_cs] (may alias y: accumulate); scale / shift fp32 [cout] device vectors or NULL (1 / 0).
 * ksplit_force: 0 = automatic (a function of the shape), >= 1 forces the split-K factor (tests). */
int w2l_convb_forward(const w2l_convb_t* c, void* stream, int N, int H, int W, const void* x, int x_cs, void* y, int y_cs,
                      const void* res, int res_cs, const float* scale, const float* shift, int ksplit_force);
/* The conv in front of a batch-statistics BatchNorm (models/conv.py:8-10,36-40 in train mode): z = conv(x) + bias (bf16, no
 * activation) AND the statistics of z in one call - mean, rstd, scale = gamma*rstd, shift = beta - mean*scale (each
 * roundup(cout,8) entries, pad entries 0) and the running-stat update, as w2l_bn_train_stats_bf16 computes them.  The sums are
 * taken from the fp32 accumulators in the conv epilogue (per-wave column partials, fixed-order reduce) whenever the launch has
 * no split-K; otherwise the stand-alone reduction over z runs.  Saves one pass over z per layer. */
int w2l_convb_forward_bn(const w2l_convb_t* c, void* stream, int N, int H, int W, const void* x, int x_cs, void* z, int z_cs,
                         const float* bias, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                         float* running_var, float* mean, float* rstd, float* scale, float* shift);
/* A data-gradient launch whose output IS the dy of a batch-statistics BatchNorm block (the block in front of this layer in
 * models/conv.py chains: wav2lip_train.py:225 loss.backward() through Conv2d -> Conv2d): y = conv(x) (+ res) in bf16 as
 * w2l_convb_forward writes it AND the two column sums BatchNorm's backward needs over that block - dbeta[c] = sum g,
 * dgamma[c] = sum g * zhat with g = y * act'(block output), zhat = (bz - mean) * rstd - taken in the conv epilogue from the
 * bf16-ROUNDED outputs (what w2l_bn_train_bwd_apply_bf16 then re-reads), per-wave column partials, fixed-order reduce.  bz:
 * the block's pre-BatchNorm conv output [N,Ho,Wo,bz_cs]; by: its output, or NULL for a ReLU block without residual (sign
 * recomputed as bz*bscale + bshift > 0); bact its activation (none / ReLU / LeakyReLU).  *fused_out = 1 when the sums were
 * written (C = roundup(cout,8) entries each, pad entries 0); 0 when the launch split K - y is written all the same and the
 * caller runs w2l_bn_train_bwd_bf16.  Saves the stand-alone reduction's pass over dy, z (and y) per block.
 * bact | W2L_BNBWD_STORE_MASKED (ReLU blocks only): when the sums are fused (*fused_out = 1) the launch stores g = dy * [block
 * output > 0] - bit for bit the bf16 dy or zero - INSTEAD of dy.  The block's own backward pass is then
 * w2l_bn_train_bwd_apply_bf16(dy = that tensor, y = NULL, act = W2L_ACT_NONE, g_out = NULL): it reads neither the block's output
 * for the mask again nor writes g (the residual path's operand is already in place): two tensor passes less per residual block.
 * With *fused_out = 0 (split-K) plain dy was stored. */
#define W2L_BNBWD_STORE_MASKED 0x100
int w2l_convb_forward_bnbwd(const w2l_convb_t* c, void* stream, int N, int H, int W, const void* x, int x_cs, void* y, int y_cs,
                            const void* res, int res_cs, const void* bz, int bz_cs, const void* by, int by_cs, int bact,
                            const float* mean, const float* rstd, const float* bscale, const float* bshift, float* dgamma,
                            float* dbeta, int* fused_out);
/* The same for an activation block WITHOUT BatchNorm (models/conv.py:22-31 `nonorm_Conv2d`: conv + LeakyReLU; the discriminator of
 * hq_wav2lip_train.py:246-256): the data-gradient launch that completes the block's dy stores dz = dy * act'(by) directly - the
 * bf16-ROUNDED dy times 1 / slope, rounded again by the store: bit for bit what w2l_act_bwd_bf16 computes from the stored dy - so
 * that the block's own backward pass needs no elementwise launch at all.  by: the block's OUTPUT; bact: ReLU or LeakyReLU.
 * dbias (optional, roundup(cout,8) floats): the column sums of the stored dz = the gradient of the block's conv bias
 * (sum over pixels), from per-wave partials of the same epilogue: the stand-alone w2l_col_sum_bf16 pass over dz is not needed.
 * *fused_out = 0 when the launch split K or ran on a special-case kernel: plain dy was stored, dbias was not written, and the
 * caller runs w2l_act_bwd_bf16 (and w2l_col_sum_bf16). */
int w2l_convb_forward_actbwd(const w2l_convb_t* c, void* stream, int N, int H, int W, const void* x, int x_cs, void* y, int y_cs,
                             const void* res, int res_cs, const void* by, int by_cs, int bact, float* dbias, int* fused_out);
/* tile override for tests / tuning: -1 = automatic */
int w2l_convb_set_tile(w2l_convb_t* c, int tile);
int w2l_convb_num_tiles(void);

/* Weight gradient on the bf16-storage path: dweight (fp32, torch layout of `g`, fully overwritten) from the layer input
 * x bf16 [N,H,W,x_cs] and the gradient dz bf16 [N,Ho,Wo,dz_cs] of the convolution output.  Same contract as w2l_conv_wgrad
 * otherwise (zero pad channels, deterministic fixed-order split-K). */
int w2l_conv_wgrad_bf16(const w2l_conv_geom* g, void* stream, int N, int H, int W, const void* x, int x_cs,
                        const void* dz, int dz_cs, float* dweight);

/* BatchNorm (batch statistics) / activation / residual passes over bf16 tensors: the bf16 twins of the fp32 entry points
 * below (same arithmetic in fp32 / fp64, 8 channels = 16 bytes per thread and row).  C = channels of the row view, a multiple
 * of 8 (pad channels included, zero in the tensors); Cvalid = channels that exist.  Per-channel OUTPUT vectors (mean, rstd,
 * scale, shift, dgamma, dbeta, column sums) have C entries - the pad entries are written as 0 so that the elementwise passes
 * can load them as vectors; per-channel INPUT parameters (gamma, beta, running stats) have Cvalid entries. */
int w2l_bn_train_stats_bf16(void* stream, long long rows, int C, int Cvalid, const void* z, int z_cs, const float* gamma,
                            const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* mean,
                            float* rstd, float* scale, float* shift);
int w2l_affine_act_bf16(void* stream, long long rows, int C, const void* z, int z_cs, const float* scale, const float* shift,
                        const void* res, int res_cs, int act, void* y, int y_cs);
/* y may be NULL for a ReLU block WITHOUT residual when `shift` (the forward's beta - mean*gamma*rstd, C entries) is given: the
 * activation mask is then recomputed as z*scale + shift > 0 - the forward's own expression - and the pass reads one tensor less */
int w2l_bn_train_bwd_bf16(void* stream, long long rows, int C, int Cvalid, const void* dy, int dy_cs, const void* y, int y_cs,
                          const void* z, int z_cs, int act, const float* mean, const float* rstd, const float* scale,
                          const float* shift, float* dgamma, float* dbeta, void* dz, int dz_cs, void* g_out, int g_cs);
/* the elementwise half of w2l_bn_train_bwd_bf16 alone, with the column sums (dgamma, dbeta: C entries) as INPUTS - after
 * w2l_convb_forward_bnbwd has produced them in the epilogue of the launch that wrote dy.  act = W2L_ACT_NONE with y = NULL: `dy`
 * already is the masked gradient g (W2L_BNBWD_STORE_MASKED). */
int w2l_bn_train_bwd_apply_bf16(void* stream, long long rows, int C, const void* dy, int dy_cs, const void* y, int y_cs,
                                const void* z, int z_cs, int act, const float* mean, const float* rstd, const float* scale,
                                const float* shift, const float* dgamma, const float* dbeta, void* dz, int dz_cs, void* g_out,
                                int g_cs);
int w2l_act_bwd_bf16(void* stream, long long rows, int C, const void* dy, int dy_cs, const void* y, int y_cs, int act,
                     const float* scale, void* dz, int dz_cs, void* g_out, int g_cs);
int w2l_add_rows_bf16(void* stream, long long rows, int C, const void* a, int a_cs, const void* b, int b_cs, void* out, int out_cs);
int w2l_col_sum_bf16(void* stream, long long rows, int C, const void* x, int x_cs, float* out);
/* The THIN 1x1 convolution (cin <= 32, cout <= 4) over [npix] rows - the generator's output layer nn.Conv2d(32, 3, 1) + Sigmoid
 * (models/wav2lip.py:83-85) in a training step: HBM-bound row kernels instead of a 128x32 GEMM tile that multiplies 29/32 padding.
 * w: fp32 [cout][cin] (the torch weight of a 1x1 conv), rounded to bf16 on the way in as the GEMM path rounds it; fp32 sums.
 * forward: y[p][o] = act(sum_c x[p][c] w[o][c] + bias[o]), 8 channels of y written (pad zero);
 * dgrad:   dx[p][c] = sum_o dz[p][o] w[o][c] (+ res[p][c]; res may alias dx), roundup(cin,8) channels written;
 * wgrad:   dweight[o][c] = sum_p dz[p][o] x[p][c] and (dbias != NULL) dbias[o] = sum_p dz[p][o], fp32, fixed summation order. */
int w2l_thin1x1_forward_bf16(void* stream, long long npix, int cin, int cout, const void* x, int x_cs, const float* w,
                             const float* bias, int act, void* y, int y_cs);
int w2l_thin1x1_dgrad_bf16(void* stream, long long npix, int cin, int cout, const void* dz, int dz_cs, const float* w,
                           const void* res, int res_cs, void* dx, int dx_cs);
int w2l_thin1x1_wgrad_bf16(void* stream, long long npix, int cin, int cout, const void* x, int x_cs, const void* dz, int dz_cs,
                           float* dweight, float* dbias);
/* graph boundary: x fp32 [N,C,H,W] -> y bf16 [N,H,W,y_cs] (channels [C, c_zero_to) zero-filled) and back */
int w2l_nchw_to_nhwc_bf16(void* stream, int N, int C, int H, int W, const float* x, void* y, int y_cs, int c_zero_to);
int w2l_nhwc_bf16_to_nchw(void* stream, int N, int C, int H, int W, const void* x, int x_cs, float* y);

/* ---------------------------------------------------------------- training: BatchNorm (batch statistics), activations
 * All tensors below are NHWC row views [rows][cs] with C valid channels; C %% 4 == 0, cs %% 4 == 0, 16-byte aligned. */

/* Per-channel batch statistics of z: mean, rstd = 1/sqrt(biased var + eps), and the affine form used by the forward
 * (scale = gamma*rstd, shift = beta - mean*scale).  running_mean / running_var (NULL = skip) are updated in place with
 * `momentum` and the unbiased variance, as nn.BatchNorm2d does in train mode. */
int w2l_bn_train_stats(void* stream, long long rows, int C, const float* z, int z_cs, const float* gamma,
                       const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                       float* mean, float* rstd, float* scale, float* shift);
/* y = act( z*scale + shift (+ res) ) */
int w2l_affine_act(void* stream, long long rows, int C, const float* z, int z_cs, const float* scale,
                   const float* shift, const float* res, int res_cs, int act, float* y, int y_cs);
/* Backward of y = act( gamma*(z-mean)*rstd + beta (+ res) ):  g = dy * act'(y);  dbeta = sum g;  dgamma = sum g*zhat;
 * dz = scale*(g - dbeta/rows - zhat*dgamma/rows) with scale = gamma*rstd.  g_out (NULL = skip; may alias dy) receives g,
 * which is also the gradient of the residual input. */
int w2l_bn_train_bwd(void* stream, long long rows, int C, const float* dy, int dy_cs, const float* y, int y_cs,
                     const float* z, int z_cs, int act, const float* mean, const float* rstd, const float* scale,
                     float* dgamma, float* dbeta, float* dz, int dz_cs, float* g_out, int g_cs);
/* g = dy * act'(y) (y may be NULL for W2L_ACT_NONE);  dz = g * scale[c] (scale NULL = 1: eval-mode BN folds to a
 * per-channel scale);  g_out as above. */
int w2l_act_bwd(void* stream, long long rows, int C, const float* dy, int dy_cs, const float* y, int y_cs, int act,
                const float* scale, float* dz, int dz_cs, float* g_out, int g_cs);
/* out = a + b (out may alias either) */
int w2l_add_rows(void* stream, long long rows, int C, const float* a, int a_cs, const float* b, int b_cs, float* out,
                 int out_cs);
/* out[c] = sum over rows of x[row][c]  (bias gradient) */
int w2l_col_sum(void* stream, long long rows, int C, const float* x, int x_cs, float* out);

/* ---------------------------------------------------------------- training: losses (gout = upstream gradient, a device
 * scalar, NULL = 1) */
int w2l_l1_mean(void* stream, long long n, const float* a, const float* b, float* loss_out);
int w2l_l1_bwd(void* stream, long long n, const float* a, const float* b, const float* gout, float* da);
int w2l_cosine_bce_bwd(void* stream, int N, int C, const float* a, const float* v, const float* y, const float* gout,
                       float* da, float* dv);
int w2l_l2norm_bwd(void* stream, int N, int C, const float* x, int x_cs, const float* dy, float* dx, int dx_cs);
int w2l_bce_bwd(void* stream, int N, const float* p, const float* y, const float* gout, float* dp);

/* ---------------------------------------------------------------- evaluation: LSE-style sync distance table
 * out [T][2*vshift+1]: out[i][j] = || f1[i] - pad(f2)[i+j] + 1e-6 ||_2 with f2 zero-padded by vshift rows on both sides
 * (calc_pdist, evaluation/scores_LSE/SyncNetInstance_calc_scores.py:19-31); f1, f2 [T][C] fp32. */
int w2l_shifted_pdist(void* stream, int T, int C, int vshift, const float* f1, const float* f2, float* out);

/* ---------------------------------------------------------------- training: fused multi-tensor Adam */
typedef struct w2l_adam_tensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    long long n;
} w2l_adam_tensor;
typedef struct w2l_adam w2l_adam_t;
/* sizes_host[ntensors]: element counts (host array); builds the block -> (tensor, chunk) table once */
int w2l_adam_create(int ntensors, const long long* sizes_host, w2l_adam_t** out);
int w2l_adam_destroy(w2l_adam_t* h);
/* One optimiser step over all tensors in ONE launch (torch.optim.Adam semantics, amsgrad off; step counts from 1).
 * tensors_host[ntensors] (host array of device pointers; gradient tensors may move between steps) must stay valid
 * until the stream has consumed the copy. */
int w2l_adam_step(w2l_adam_t* h, void* stream, const w2l_adam_tensor* tensors_host, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int step);

/* ---------------------------------------------------------------- plans (a recorded sequence of launches) */

typedef struct w2l_plan w2l_plan_t;
int w2l_plan_create(w2l_plan_t** out);
int w2l_plan_destroy(w2l_plan_t* p);
/* record one conv launch with fixed buffers/shapes; replayed in order by w2l_plan_run */
int w2l_plan_add_conv(w2l_plan_t* p, const w2l_conv_t* c, int N, int H, int W, const float* x, int x_cs,
                      float* y, int y_cs, const float* res, int res_cs);
/* append a copy of launch `index` of `src` (same layer, buffers and configuration) to `dst`: sub-plans for multi-stream runs */
int w2l_plan_copy_item(w2l_plan_t* dst, const w2l_plan_t* src, int index);
int w2l_plan_run(const w2l_plan_t* p, void* stream);
int w2l_plan_size(const w2l_plan_t* p);
/* OPT-IN stopwatch autotune: time every (tile configuration, split-K factor) candidate of every recorded launch on its real
 * buffers (reps timed runs each, HIP events on `stream`, synchronises), keep the fastest for w2l_plan_run and record it in
 * the tune table below.  A stopwatch choice differs from box to box and run to run, and with it the summation order of the
 * layer: the default path (no autotune) is the table, then a heuristic - both functions of the shape only. */
int w2l_plan_autotune(w2l_plan_t* p, void* stream, int reps);
int w2l_plan_get_config(const w2l_plan_t* p, int index, int* tile, int* ksplit);
int w2l_plan_set_config(w2l_plan_t* p, int index, int tile, int ksplit);   /* tile -1 = heuristic */
/* FLOPs the matrix cores EXECUTE per recorded launch with its current configuration (padded tiles and K; Winograd layers:
 * 16 products per 2x2 output tile per (cin, cout) instead of 36): flops_out[w2l_plan_size].  Launches nothing.  This is the
 * numerator of bench.py's roofline fraction (the nominal direct-convolution count is w2l_conv_macs).  config_out (optional,
 * [w2l_plan_size][2]) receives the (configuration id, split-K) each launch resolves to - explicit, tune table or heuristic;
 * ids below w2l_conv_num_igemm_tiles() are implicit-GEMM tiles, the rest Winograd configurations. */
int w2l_plan_executed_flops(const w2l_plan_t* p, long long* flops_out, int* config_out);
int w2l_conv_num_igemm_tiles(void);
/* Executed-FLOP counter for whole training steps (tools/train_bench.py): between w2l_flops_begin() and w2l_flops_end() every
 * conv / data-gradient / weight-gradient launch of the process (any thread: backward() runs on torch's autograd thread) ALSO adds the multiply-add work its matrix cores execute
 * (padded tiles and K, 16 or 9 instead of 36 products on Winograd launches) to a counter; w2l_flops_end returns the total and,
 * if by_family != NULL, by_family[8]: 0 fp32 conv, 1 fp32 weight gradient (direct), 2 fp32 Winograd weight gradient, 3 / 4 the
 * round-2 bf16-contraction conv / weight gradient, 5 / 6 the bf16-storage conv / weight gradient, 7 non-MFMA head reductions. */
int w2l_flops_begin(void);
long long w2l_flops_end(long long* by_family);
/* Shader-clock probe (measurement aid of bench.py, no reference counterpart): one wave spins `spin_us` microseconds on `stream`
 * and writes out2_dev[0] = shader-clock ticks (s_memtime), out2_dev[1] = 100 MHz reference ticks (s_memrealtime) of the spin;
 * MHz = 100 * out[0] / out[1].  Run on a side stream while a workload runs: the clock the chip sustains under that workload. */
int w2l_clock_probe(void* stream, int spin_us, unsigned long long* out2_dev);
/* kernel family of a configuration id: 0 = conv_igemm_f32_kernel, 1 = conv_wino_f32_kernel, 2 = conv_wino2_f32_kernel,
 * (conv_wino2.hip: ids 8, 9 and the quarter-split shape, id 12), 3 = conv_tp2_f32_kernel (stride-2 transposed 3x3, all four
 * phases in one workgroup), 4 = conv_wino4_f32_kernel (Winograd F(4x4,3x3)), 5 = the split-operand implicit GEMM (ids 13..18 =
 * the implicit-GEMM tiles 0..5 again, W2L_PREC_F32 layers only: every fp32 operand enters the bf16 matrix cores as the exact sum of
 * three bf16 pieces, six piece products per product, fp32 accumulate - an fp32 result with the fp32 kernels' error, not bitwise
 * theirs; the committed table gives it ten batch-128 generator launches, W2L_SPLIT=0 puts their fp32-pipe entries back), 6 = the
 * same arithmetic inside Winograd F(2x2,3x3) (conv_wino2s_kernel, id 19), 7 = inside the fused-phase stride-2 transposed kernel
 * (conv_tp2s_kernel, id 20), 8 = the generator's 7x7 first layer with the region staged and split once (conv_stem7s_kernel, id 21),
 * 9 = the direct 3x3 kernel for 32-cout layers with the optional fused 1x1 head (conv_k3s_kernel, id 22); -1 = bad id.  Ids are append-only across library versions. */
int w2l_conv_config_family(int id);
/* Switch kernel families off (bit f of `mask` = family f of w2l_conv_config_family; family 0, the implicit GEMM, cannot be
 * excluded): w2l_plan_autotune skips their ids and a table / forced id of an excluded family falls through to the next rule.
 * mask < 1024.  mask 16 (= no conv_wino4) is how the "exact" launch table is built and run: F(4x4,3x3) carries about twice the rounding
 * error of F(2x2,3x3) (7.7e-7 vs 4.2e-7 pixel L-inf against the reference). */
int w2l_conv_exclude_families(int mask);
/* The implicit-GEMM kernels' workgroup -> (phase, M-tile, cout-tile) map (measurement / test aid, no reference counterpart; host
 * code, runs without a GPU): out[3 * i ..] = (phase, tile_m, tile_n) of the i-th workgroup in dispatch order for a grid of
 * tiles_m * tiles_n x nphase workgroups (every K-split repeats it); workgroup i runs on XCD i % 8.  order 0 = phase-major, cout-tiles
 * fastest inside a phase; 1 = phase-major, cout-tile slowest inside a phase (weight-heavy layers: an XCD fetches its 1/8 of the
 * weights once); 2 = groups of order_r M-tiles phase by phase (a stride-2 transposed layer's four phases re-read their input out of an
 * XCD's L2 instead of HBM).  The launcher chooses per shape. */
int w2l_igemm_block_order(int order, int order_r, int tiles_m, int tiles_n, int nphase, int* out);
/* time each recorded launch with HIP events on `stream` (reps runs, averaged): ms_out[w2l_plan_size] */
int w2l_plan_profile(const w2l_plan_t* p, void* stream, int reps, float* ms_out);

/* ---------------------------------------------------------------- tune table (shape -> launch configuration)
 * A conv launch without an explicit configuration looks its shape up here; a miss runs the library's heuristic.  Both are
 * pure functions of the shape, so results are bit-reproducible across runs and boxes (the reference's torch ops are
 * deterministic on CPU; a stopwatch-tuned path is not).  Key = W2L_TUNE_KEY_INTS ints: transposed, cin, cout, kh, kw, sh,
 * sw, ph, pw, oph, opw, precision (W2L_PREC_*), has_residual, head_c, N, H, W.  The host side loads the committed
 * wav2lip_amd/tune_table.json at start-up (tools/make_tune_table.py regenerates it on a GPU box). */
#define W2L_TUNE_KEY_INTS 17
int w2l_tune_key_ints(void);
int w2l_tune_set(const int* key, int tile, int ksplit);
/* 1 if configuration id `tile` can run a launch with this key (geometry / precision / residual / head eligibility of the
 * kernel family behind the id), 0 if such a launch would fall through to the heuristic: table writers and the table test use it */
int w2l_tune_entry_applicable(const int* key, int tile);
int w2l_tune_clear(void);
int w2l_tune_count(void);
/* out[cap_entries][W2L_TUNE_KEY_INTS + 2] = key, tile, ksplit per entry; returns the number of entries written */
int w2l_tune_export(int* out, int cap_entries);

#ifdef __cplusplus
}
#endif
#endif /* W2L_HIP_H */
