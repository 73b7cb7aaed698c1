/*
 * gsa_train.h -- C ABI of the decoder-training operators.
 *
 * The reference trains the decoder with gluon autograd over MXNet operators (seg_solver.py:351-465:
 * `self.net(*features)` in train mode, SoftmaxCELoss with sample weights, `err.backward()`, Adam via
 * `trainer.step`).  Here each operator (forward and backward) is one HIP entry point and the graph is written
 * out by hand in gan-segmentation_amd/trainer.py -- the role gluon's Python layer plays in the reference.
 *
 * Conventions: every pointer is a DEVICE pointer on the current HIP device unless stated; tensors are fp32
 * NCHW (the reference's layout: features come from `feat_*.pickle` as CHW arrays, weights are OIHW); `stream`
 * is a hipStream_t as void*; calls are stream-ordered and never synchronise; 0 on success, a negative
 * gsa_status otherwise.  Unlike the inference path these kernels are not order-canonical: sums use float
 * atomics, so results are reproducible to fp32 rounding, not bit for bit; parity is a stated tolerance against
 * a torch-autograd restatement (oracle/ref_train.py).  The two BatchNorm operators reduce through one per-device
 * scratch buffer: issue them from one stream at a time (the trainer uses a single stream).
 */
#ifndef GSA_TRAIN_H
#define GSA_TRAIN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Convolution K x K (K = 3: pad 1, K = 1: pad 0), stride 1, over the channel concatenation of two sources
 * (x1 may be NULL) that are optionally nearest-upsampled x2 on read (UpSample, networks_stylegan.py:308-315):
 *   out[n][o][y][x] = bias[o] + sum_{c,ky,kx} in[n][c][y+ky-p][x+kx-p] * W(o,c,ky,kx),   H = Hs << up.
 * transposed = 0: W(o,c,ky,kx) = w[o][c][ky][kx], w shaped (Cout, C0+C1, K, K)          -- forward (nn.Conv2D)
 * transposed = 1: W(o,c,ky,kx) = w[c][o][K-1-ky][K-1-kx], w shaped (C0+C1, Cout, K, K)  -- gradient w.r.t. the
 *                 input of the forward conv whose weight is w (in = dL/dout of that conv).
 * The Cout output channels are split over two destinations: channels [0, Cout0) go to out0 (n,Cout0,H,W), the
 * rest to out1 (n,Cout-Cout0,H,W) -- the mirror of the two-source concat; out1 may be NULL when Cout0 == Cout.
 * accumulate != 0 adds to the destinations instead of overwriting them.  bias may be NULL. */
int gsa_train_conv(void* stream, int32_t n, const float* x0, int32_t C0, const float* x1, int32_t C1, int32_t Hs,
                   int32_t Ws, int32_t up, const float* w, int32_t Cout, int32_t K, int32_t transposed,
                   const float* bias, float* out0, int32_t Cout0, float* out1, int32_t accumulate);

/* Gradient of the same convolution w.r.t. its weight and bias:
 *   dw[o][c][ky][kx] += sum_{n,y,x} dy[n][o][y][x] * in[n][c][y+ky-p][x+kx-p],   db[o] += sum dy[n][o][y][x]
 * (in = the forward input, concat / upsample as above; dy (n,Cout,H,W); db may be NULL).  Float atomics. */
int gsa_train_conv_wgrad(void* stream, int32_t n, const float* x0, int32_t C0, const float* x1, int32_t C1,
                         int32_t Hs, int32_t Ws, int32_t up, const float* dy, int32_t Cout, int32_t K, float* dw,
                         float* db);

/* BatchNorm in training mode (nn.BatchNorm: eps 1e-5, momentum 0.9, batch statistics over N,H,W, biased
 * variance) followed by LeakyReLU(0.2) and an optional Dropout mask (networks_seg.py:14-32,64-79):
 *   mean/var (C each) = batch statistics of v;   running = momentum*running + (1-momentum)*batch;
 *   y = lrelu(gamma*(v-mean)/sqrt(var+eps) + beta) * (mask ? mask*drop_scale : 1).
 * mask: u8 (n,C,HW) keep flags or NULL. */
int gsa_train_bn_lrelu_fwd(void* stream, int32_t n, int32_t C, int32_t HW, const float* v, const float* gamma,
                           const float* beta, float eps, float momentum, float* mean, float* var,
                           float* running_mean, float* running_var, const uint8_t* mask, float drop_scale, float* y);

/* Backward of the above: g holds dL/dy on entry and dL/dv on exit; dgamma/dbeta (C each) are accumulated. */
int gsa_train_bn_lrelu_bwd(void* stream, int32_t n, int32_t C, int32_t HW, const float* v, const float* gamma,
                           const float* beta, float eps, const float* mean, const float* var, const uint8_t* mask,
                           float drop_scale, float* g, float* dgamma, float* dbeta);

/* SyncBatchNorm (gluon.contrib.nn.SyncBatchNorm when cfg['use_sync_bn'], networks_seg.py:20-21,30-31,73-74): the two calls
 * above cut at the per-channel sums, so that the host can all-reduce them over the ranks (RCCL, 2*C doubles) in between.
 *   gsa_train_bn_sums:            sums[c] = sum v, sums[C+c] = sum v*v over this rank's (n, HW)
 *   gsa_train_bn_lrelu_fwd_sums:  mean/var/running/y as gsa_train_bn_lrelu_fwd, from `sums` over `count` values per channel
 *                                 (count = the pixels of ALL ranks, >= n*HW)
 *   gsa_train_bn_bwd_sums:        sums[c] = sum dz, sums[C+c] = sum dz*xhat of this rank (dz = dL/dy through dropout and LeakyReLU)
 *   gsa_train_bn_lrelu_bwd_sums:  g <- dL/dv with the batch means taken from sums_all / count; dgamma/dbeta += sums_own (this
 *                                 rank's share: the gradient all-reduce adds the ranks up)
 * With one rank (sums_all = sums_own, count = n*HW) the results equal the fused calls bit for bit. */
int gsa_train_bn_sums(void* stream, int32_t n, int32_t C, int32_t HW, const float* v, double* sums);
int gsa_train_bn_lrelu_fwd_sums(void* stream, int32_t n, int32_t C, int32_t HW, double count, const float* v, const float* gamma,
                                const float* beta, float eps, float momentum, const double* sums, float* mean, float* var,
                                float* running_mean, float* running_var, const uint8_t* mask, float drop_scale, float* y);
int gsa_train_bn_bwd_sums(void* stream, int32_t n, int32_t C, int32_t HW, const float* v, const float* gamma, const float* beta,
                          float eps, const float* mean, const float* var, const uint8_t* mask, float drop_scale, const float* g,
                          double* sums);
int gsa_train_bn_lrelu_bwd_sums(void* stream, int32_t n, int32_t C, int32_t HW, double count, const float* v, const float* gamma,
                                const float* beta, float eps, const float* mean, const float* var, const uint8_t* mask,
                                float drop_scale, const double* sums_all, const double* sums_own, float* g, float* dgamma,
                                float* dbeta);

/* SoftmaxCELoss(axis=1) with sample weight 1 on labelled pixels and 0 on ignored ones (label -1), per-sample mean
 * over H*W (seg_solver.py:243-250, 395-407): loss[n] and dlogits = grad_scale * d(sum_n loss[n])/dlogits. */
int gsa_train_softmax_ce(void* stream, int32_t n, int32_t classes, int32_t HW, const float* logits,
                         const int8_t* labels, float* loss, float* dlogits, float grad_scale);

/* dx[n][c][y][x] = sum of the 2x2 block of dy_up (backward of nearest x2).  accumulate as above. */
int gsa_train_upsample2_bwd(void* stream, int32_t n, int32_t C, int32_t Hs, int32_t Ws, const float* dy_up, float* dx,
                            int32_t accumulate);

/* out = a + b (residual add and gradient accumulation), count elements; out may alias a. */
int gsa_train_add(void* stream, int64_t count, const float* a, const float* b, float* out);

/* Dropout keep mask: mask[i] = uniform(seed, stream_id, i) < keep_prob, Philox4x32-10 counter = (i/4, stream_id). */
int gsa_train_dropout_mask(void* stream, int64_t count, uint64_t seed, uint32_t stream_id, float keep_prob, uint8_t* mask);

/* MXNet's Adam update (optimizer 'adam', seg_solver.py:203-219):  g' = g*rescale + wd*w;  m = b1*m+(1-b1)*g';
 * v = b2*v+(1-b2)*g'^2;  w -= lr_t * m / (sqrt(v) + eps)   with lr_t = lr*sqrt(1-b2^t)/(1-b1^t) from the caller. */
int gsa_train_adam(void* stream, int64_t count, float* w, const float* g, float* m, float* v, float lr_t, float beta1,
                   float beta2, float eps, float rescale, float wd);

#ifdef __cplusplus
}
#endif
#endif /* GSA_TRAIN_H */
