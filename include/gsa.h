/*
 * gsa.h -- C ABI of the MI355X-native `generate` hot path
 * (StyleGAN-v1 synthesis + segmentation decoder -> (image, mask) pairs).
 *
 * The reference has no FFI of its own: its boundary is a Python call surface over
 * MXNet NDArrays.  Each entry point below names the reference
 * interface it replaces; the ctypes stub a maintainer of the reference would add is
 * shown in INTEGRATION.md.
 *
 * Conventions
 *  - every function returns 0 on success or a negative gsa_status; it never throws and
 *    never synchronises the stream unless stated;
 *  - the caller owns every buffer; pointers marked `dev` are device (HBM) pointers valid
 *    on the context's device, pointers marked `host` are host pointers;
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream); all work of a
 *    call is enqueued on it in order;
 *  - one context per (device, host thread); contexts are not thread-safe;
 *  - instance-norm statistics are order-independent 64-bit fixed-point sums per aligned quad of 4 pixels: sums in units of
 *    2^-28, sums of squares in units of 2^-S2 with S2 = clamp(40 - ceil(log2(H*W)), 20, 26), a static function of the plane
 *    size (20 at 1024^2, 22 / 24 at 512^2 / 256^2, 26 from 128^2 down; a quad whose sum of squares is >= 2^(50-S2) is rounded at
 *    2^-20 and shifted into the unit).  Both ends of the range:
 *      upper: exact and deterministic while |x| < 2.3e4 per value and rms(x) < 2.9e3 over a plane (any size); beyond that the
 *             sums wrap.  A sum within a factor 4 of the wrap (or a variance that comes out negative) sets a sticky device word
 *             that gsa_check (and, in flight, gsa_status_snapshot) reports as GSA_ERR_DEVICE;
 *      lower: the variance E[x^2] - mean^2 carries an absolute error of at most 2^-(S2+3) (all quads rounding the same way;
 *             typically 2^-(S2+4) / sqrt(quads)): <= 1.9e-9 on a 4x4 plane, i.e. <= 2e-4 of the instance norm's eps 1e-5, so a
 *             plane of ANY spread -- down to a constant plane, and a spread far below its mean (values riding on a bias) --
 *             is normalised to within the 1e-3 tolerance of the reference's two-pass form (tests: a level's weights scaled by
 *             1e-3 / 1e-4 and by 800 against oracle/ref_semantic.py).  Post-LeakyReLU StyleGAN-v1 activations are O(1..100);
 *  - tensors crossing the boundary use the reference's layouts: fp32 NCHW activations,
 *    OIHW conv weights, (N,H,W,3) u8 RGB images, (N,H,W) u8 masks.
 *
 * The same signatures with the prefix `gsao_` are exported by the CPU oracle
 * (oracle/c/gsa_oracle.c, test infrastructure only) with every pointer a host pointer.
 */
#ifndef GSA_H
#define GSA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gsa_ctx gsa_ctx;

typedef enum {
    GSA_OK = 0,
    GSA_ERR_INVALID = -1,      /* bad argument / shape / unsupported configuration */
    GSA_ERR_STATE = -2,        /* call order (e.g. forward before commit) */
    GSA_ERR_MISSING_PARAM = -3,/* a declared parameter was never set (reference: load_parameters without allow_missing) */
    GSA_ERR_HIP = -4,          /* HIP runtime error; see gsa_last_error */
    GSA_ERR_NOMEM = -5,
    GSA_ERR_DEVICE = -6        /* a device-side check failed (reported by gsa_check: see there) */
} gsa_status;

/* Generator(config): reference networks_stylegan.py:78-91 (+ image_generator.py:46-74). */
typedef struct {
    int32_t max_res_log2;   /* 10 ffhq, 9 cars, 8 bedrooms (reference image_generator.py:11) */
    int32_t fmap_base;      /* 8192 */
    double  fmap_decay;     /* 1.0 */
    int32_t fmap_max;       /* 512 */
    int32_t latent_size;    /* 512 */
    int32_t channels;       /* 3 */
    int32_t use_wscale;     /* 1 */
} gsa_generator_config;

/* Decoder(cfg): reference networks_seg.py:51-62, seg_solver.py:119-128. */
typedef struct {
    int32_t num_feats;            /* len(in_channels) */
    int32_t start_res;            /* first feature consumed, 0..num_feats-1 (networks_seg.py:56; the reference sets 0) */
    int32_t use_bn;               /* 1 */
    const int32_t* features;      /* host, num_feats+1 entries; last = num_classes */
    const int32_t* in_channels;   /* host, num_feats entries */
} gsa_decoder_config;

/* Context on HIP device `device`.  Replaces the implicit MXNet context list
 * (reference image_generator.py:17, seg_solver.py:24-28). */
int gsa_create(int device, gsa_ctx** out);
void gsa_destroy(gsa_ctx* ctx);

/* Message of the last failing call on `ctx` (or of the last failing gsa_create when ctx is
 * NULL).  The reference raises Python exceptions; the Python shim turns this into one. */
const char* gsa_last_error(const gsa_ctx* ctx);

/* Generator.__init__ (reference networks_stylegan.py:78-112). */
int gsa_generator_init(gsa_ctx* ctx, const gsa_generator_config* cfg);

/* One tensor of `Generator.load_parameters` (reference image_generator.py:21-22), addressed
 * by its scheme-P name, fp32, the reference's shape and layout.
 * Unknown names are ignored (ignore_extra=True) and reported as 1 (not an error). */
int gsa_generator_set_param(gsa_ctx* ctx, const char* name, const float* host_data,
                            int32_t ndim, const int64_t* dims);

/* Finish loading: every declared parameter must have been set (no allow_missing in the
 * reference).  Computes the effective weights (W*std)*lr_mult (reference
 * networks_stylegan.py:407-412,513-518), repacks them for the kernels and uploads. */
int gsa_generator_commit(gsa_ctx* ctx);

/* Decoder.__init__ / load_parameters (reference networks_seg.py:51-94, seg_solver.py:339-349);
 * names are the structural names the reference itself writes. */
int gsa_decoder_init(gsa_ctx* ctx, const gsa_decoder_config* cfg);
int gsa_decoder_set_param(gsa_ctx* ctx, const char* name, const float* host_data,
                          int32_t ndim, const int64_t* dims);
int gsa_decoder_commit(gsa_ctx* ctx);

/* Size the activation workspace for batches up to `max_batch` samples per call.
 * (Allocation happens here, never inside a forward call.) */
int gsa_reserve(gsa_ctx* ctx, int32_t max_batch);

/* Generator.hybrid_forward (reference networks_stylegan.py:165-197) + _transform_gan_back
 * (reference image_generator.py:76-84).
 *   z      dev (N, latent_size) fp32
 *   noise  host array of 2*(max_res_log2-1) dev pointers; plane l is (N,1,R,R) fp32 with
 *          R = 4,4,8,8,...: the N(0,1) draws of AddNoise (reference networks_stylegan.py:297-300)
 *   rgb    dev (N,channels,R,R) fp32 or NULL     -- first return value of the reference
 *   img    dev (N,R,R,channels) u8 or NULL       -- _transform_gan_back of rgb
 *   feats  NULL or host array of max_res_log2-1 dev pointers, each (N,C_r,R_r,R_r) fp32 or
 *          NULL                                   -- second return value of the reference
 *   num_noise / num_feats  entries in the two pointer arrays: must equal 2*(max_res_log2-1) and (when feats is
 *          given) max_res_log2-1 of THIS context's generator -- a caller holding arrays sized for another
 *          configuration gets GSA_ERR_INVALID instead of an out-of-bounds read */
int gsa_generator_forward(gsa_ctx* ctx, void* stream, int32_t n, const float* z,
                          const float* const* noise, int32_t num_noise, float* rgb, uint8_t* img,
                          float* const* feats, int32_t num_feats);

/* Decoder.hybrid_forward + argmax of SegSolver.predict (reference networks_seg.py:97-113,
 * seg_solver.py:321-327).
 *   feats  host array of num_feats dev pointers, each (N,in_channels[i],R_i,R_i) fp32
 *   logits dev (N,num_classes,R,R) fp32 or NULL
 *   num_feats  entries in feats; must equal the decoder's num_feats (GSA_ERR_INVALID otherwise)
 *   mask   dev (N,R,R) u8 class index (first maximum wins) or NULL */
int gsa_decoder_forward(gsa_ctx* ctx, void* stream, int32_t n, const float* const* feats,
                        int32_t num_feats, float* logits, uint8_t* mask);

/* The fused `main.py generate` step (reference main.py:97-99): generator and decoder with
 * the feature maps kept in HBM in the kernels' own layout.  Bitwise identical to
 * gsa_generator_forward + gsa_decoder_forward on the same inputs. */
int gsa_generate(gsa_ctx* ctx, void* stream, int32_t n, const float* z,
                 const float* const* noise, int32_t num_noise, uint8_t* img, uint8_t* mask);

/* gsa_generate runs the decoder on a second HIP stream beside the synthesis of the higher
 * resolutions (fork/join through events).  `levels` = number of decoder levels placed there; negative = the default:
 * all but the last (measured faster at every batch size with the round-2 kernels);
 * 0 = everything on the caller's stream, used by bench.py's serialized roofline pass so that kernel durations are
 * not stretched by concurrent kernels. */
int gsa_set_overlap(gsa_ctx* ctx, int32_t levels);

/* Arithmetic of the MFMA convolutions (BASELINE.json configs[4], "bf16 MFMA"; the reference itself is fp32
 * only, networks_stylegan.py / networks_seg.py use MXNet's default dtype).
 *   GSA_PREC_F32  (default) exact-fp32 MFMA, the canonical bit-exact path;
 *   GSA_PREC_BF16 conv / deconv inputs (after the fp32 AdaIN) and weights are rounded to bf16 (RNE) and
 *                 multiplied on v_mfma_f32_16x16x16_bf16; accumulation, bias/BN/LeakyReLU, noise, instance-norm
 *                 statistics, mapping network, toRGB and the final 32->classes conv stay fp32.
 * Must be called before the commits (weights are packed per mode): GSA_ERR_STATE otherwise. */
typedef enum { GSA_PREC_F32 = 0, GSA_PREC_BF16 = 1 } gsa_precision;
int gsa_set_precision(gsa_ctx* ctx, int32_t mode);

/* Counter-based N(0,1) inputs: z (N,latent_size) and/or the 2*(max_res_log2-1) noise planes (N,1,R,R), each
 * element a pure function of (seed, first_index + sample, plane, element) -- Philox4x32-10 + Box-Muller -- so a
 * sample's inputs do not depend on the batch, rank or GPU count that produces it.
 * Replaces mx.nd.random.randn (reference image_generator.py:94) and AddNoise's random_normal
 * (networks_stylegan.py:297-300) when the caller wants reproducible shards; latent_size % 4 == 0.
 * Either pointer may be NULL; num_noise = entries in noise (must be 2*(max_res_log2-1) when noise is given). */
int gsa_fill_inputs(gsa_ctx* ctx, void* stream, int32_t n, uint64_t seed, uint64_t first_index, float* z,
                    float* const* noise, int32_t num_noise);

/* Per-batch arithmetic of SegSolver.evaluate_for_data (reference seg_solver.py:229-262) and
 * SegmentationMetric.update (reference metrics.py:497-606), SURVEY.md section 8f-4:
 *   logits     dev (N,classes,H,W) fp32 (gsa_decoder_forward's logits)
 *   labels     dev (N,H,W) int8: class index, -1 = ignore (seg_datasets.py:85-106)
 *   confusion  dev [classes*classes] u64, ACCUMULATED: confusion[l*classes+p] += #pixels with label l >= 0 and
 *              argmax p (first maximum).  pixAcc / IoU follow from it: correct = trace, labelled = sum,
 *              inter = diagonal, union = row sum + column sum - diagonal.
 *   loss_fixed dev [N] u64, ACCUMULATED: sum over the sample's labelled pixels of rint(err * 2^32) with
 *              err = -log_softmax(logits)[label] in fp32 (SoftmaxCELoss with sample weight 1 on labelled
 *              pixels, 0 on ignored ones); the per-sample loss is loss_fixed / 2^32 / (H*W).
 * Both buffers are caller-zeroed.  classes 2..8. */
int gsa_segmentation_eval(gsa_ctx* ctx, void* stream, int32_t n, int32_t classes, int32_t H, int32_t W,
                          const float* logits, const int8_t* labels, uint64_t* confusion, uint64_t* loss_fixed);

/* Device-side checks.  The forward calls are stream-ordered and return before their kernels ran, so two conditions that a
 * kernel can only detect on the device are recorded in sticky device words and reported HERE:
 *   - the fused mapping network (PixelNorm + 8 DenseW in one launch, reference networks_stylegan.py:128-139) exchanges
 *     activations between its workgroups; if a workgroup waited longer than ~0.25 s for a partner (the launch's workgroups
 *     were not co-resident: CU masking, a long kernel of another stream holding the chip) the latents of that step are invalid;
 *   - an instance-norm statistic (networks_stylegan.py:246,261) came within a factor 4 of the 64-bit wrap of its fixed-point
 *     sum, or produced a negative variance: activations beyond the range stated above.
 * gsa_check synchronises the device, returns GSA_OK or GSA_ERR_DEVICE (gsa_last_error names the condition) and clears the
 * words.  The Python shim calls it after a context's first step and when a context is closed; gsa_reserve and
 * gsa_profile_collect (which synchronise anyway) report the same condition.  Results produced since the previous clean check
 * must be discarded when it fails. */
int gsa_check(gsa_ctx* ctx);

/* The same two words WITHOUT a synchronisation, for a long run in flight (main.py generate: 10 000 samples, reference
 * main.py:93-104): enqueues an 8-byte copy of {statistics-range word, mapping time-out word} to `host_words` (pinned host
 * memory, 2 x uint32) on `stream`, behind the kernels already enqueued there.  The words are sticky, so a snapshot taken
 * after batch k that reads {0, 0} proves every batch up to k clean; the first non-zero snapshot names the first bad batch.
 * The caller reads the words once an event recorded behind this call has completed (the dataset writer does, before it
 * releases that batch's files).  Does not clear the words: gsa_check does. */
int gsa_status_snapshot(gsa_ctx* ctx, void* stream, uint32_t* host_words);

/* Test hook (the product never calls it) -- doubly gated: it arms something only when THIS call is made AND the process runs with
 * GSA_TEST_HOOKS=1 (tests/conftest.py sets it); without the variable the call returns GSA_ERR_STATE, and the variable alone arms
 * nothing:
 *   kind 0  disarm everything;
 *   kind 1  the fused mapping network is launched one workgroup short (its exchange times out: GSA_ERR_DEVICE at the check);
 *   kind 2  the NEXT generator pass returns GSA_ERR_HIP between a statistics producer and its finalize (dirty rows);
 *   kind 3  `arg` generator passes from now the statistics-range word is set on that pass's stream, as if an instance-norm
 *           sum had left its range there (arg = 0: the next pass). */
int gsa_debug_inject(gsa_ctx* ctx, int32_t kind, int32_t arg);

/* --- measurement hooks (bench.py) ------------------------------------------------------ */

/* When enabled every kernel launch is bracketed by hipEvents on the launch stream. */
int gsa_profile_enable(gsa_ctx* ctx, int32_t on);
/* Synchronises the recorded events and returns the number of distinct kernel labels. */
int gsa_profile_collect(gsa_ctx* ctx);
/* Label i: name, accumulated milliseconds, launch count, and summed over those launches: the FLOP the kernel executes,
 * its algorithmic bytes, and the FLOP of the same layers in the reference's formulation (2*MACs of the direct
 * convolution, SURVEY.md section 8d -- larger than `flops` where the kernel uses the sub-pixel or the Winograd form).
 * Any out pointer may be NULL.  Returns 0, or GSA_ERR_INVALID when i is out of range. */
int gsa_profile_entry(gsa_ctx* ctx, int32_t i, const char** name, double* ms, int64_t* launches,
                      double* flops, double* bytes, double* alg_flops);
int gsa_profile_reset(gsa_ctx* ctx);

/* Library build string: version, arch, and the compiler it was built with (the hand-placed s_nop hazard padding around the
 * inline-asm packed adds of the Winograd kernels was validated on exactly that compiler: DESIGN.md section 4). */
const char* gsa_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GSA_H */
