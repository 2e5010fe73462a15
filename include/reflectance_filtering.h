/*
 * reflectance_filtering.h -- C ABI of librf_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the three native calls on the reference's hot path.
 * Every entry point takes plain device pointers and sizes, runs asynchronously
 * on the HIP stream it is given (hipStream_t passed as void*; NULL = the null
 * stream), never allocates or synchronises on the launch path once its
 * parameter tables are cached, and never throws: it returns RF_OK or a negative
 * RF_E* code, and rf_last_error() returns the calling thread's message.
 *
 * There is no CPU fallback in this library.  Images are uint8, interleaved
 * (H x W x C), tightly packed, batched along the leading dimension n; every
 * image of a batch is filtered independently (no inter-image or inter-GPU
 * traffic; shard batches across GPUs by giving each process its own slice).
 */
#ifndef REFLECTANCE_FILTERING_H
#define REFLECTANCE_FILTERING_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RF_VERSION 103 /* 0.1.3 */

/* return codes */
#define RF_OK 0
#define RF_E_BADARG (-1)      /* NULL pointer, non-positive size, sigma handling is OpenCV's (<=0 -> 1) */
#define RF_E_UNSUPPORTED (-2) /* channel count / radius / border outside what is implemented */
#define RF_E_WORKSPACE (-3)   /* workspace too small for one image */
#define RF_E_HIP (-4)         /* a HIP runtime call failed; see rf_last_error() */

/* border types, numerically equal to cv::BorderTypes */
#define RF_BORDER_CONSTANT 0
#define RF_BORDER_REPLICATE 1
#define RF_BORDER_REFLECT 2
#define RF_BORDER_WRAP 3
#define RF_BORDER_REFLECT_101 4
#define RF_BORDER_DEFAULT RF_BORDER_REFLECT_101

/* rf_jbf_u8 flags */
#define RF_JBF_TRUE_DIVISION 1 /* dst = sum / wsum; default is OpenCV's sum * (1.f / wsum) */
#define RF_JBF_FORCE_GENERIC 2 /* use the untiled global-memory kernel (debug / cross-check) */
#define RF_JBF_GREY_AS_BGR 4    /* a 1-channel joint counts as 3 equal channels, i.e. what cv2.imread makes of a
                                 grey PNG (colour distance 3*|d|); with a 1-channel src the 1-channel result
                                 equals every channel of the 3-channel one */
/* any other flag bit is rejected (RF_E_BADARG); test and benchmark switches live in
   reflectance_filtering_debug.h */

int rf_version(void);
const char *rf_last_error(void);
/* Frees what the library holds: the per-device parameter tables (colour LUTs, tap tables), the
   packed-weight slots of rf_cnn_reflectance_u8 and the guided filter's side streams.  Call it with
   no library work in flight. */
int rf_shutdown(void);

/*
 * Joint bilateral filter, 8-bit.
 * Replaces  cv2.ximgproc.jointBilateralFilter(joint, src, d, sigmaColor, sigmaSpace[, borderType])
 * as called at /root/reference/filter_reflectance.py:60-64 (d = -1).
 *   joint  n*h*w*joint_cn   device, joint_cn in {1,3}
 *   src    n*h*w*src_cn     device, src_cn   in {1,3}
 *   dst    n*h*w*src_cn     device, must not overlap joint or src
 *   d <= 0 -> radius = cvRound(1.5*sigma_space), else radius = d/2; radius >= 1
 * Per pixel the taps are accumulated in OpenCV's order (row-major over the
 * disk, float32, separately rounded multiply and add), so the uint8 result is
 * bit-identical to the restated CPU algorithm.
 */
int rf_jbf_u8(const uint8_t *joint, const uint8_t *src, uint8_t *dst, int n, int h, int w,
              int joint_cn, int src_cn, int d, double sigma_color, double sigma_space, int border,
              int flags, void *stream);

/*
 * Guided filter, 8-bit, colour guide.
 * Replaces  cv2.ximgproc.guidedFilter(guide, src, radius, eps)  as called at
 * /root/reference/filter_reflectance.py:67-70 (radius = int(sigma_spatial), eps = sigma_color).
 *   guide  n*h*w*3        device (guide_cn must be 3: cv2.imread always yields 3 channels)
 *   src    n*h*w*src_cn   device, src_cn in {1,3}
 *   dst    n*h*w*src_cn   device; may alias src
 *   radius 0..4096 (int(sigma_spatial) is a free parameter of the reference's tool): radii up to
 *   120 run the 8-bit kernels (exact uint32 window sums); larger ones run the float kernels of
 *   rf_gf_f32 on float copies of the images kept in the workspace, each pass rounded to uint8 like
 *   convertTo(CV_8U) - the same bytes (on 8-bit data the float path's double window sums are the
 *   same exact integers).  rf_gf_workspace_bytes sizes the workspace for the radius it is given.
 *   iterations >= 1: the filter is applied `iterations` times with the same guide, the uint8
 *   result of one pass being the src of the next (the reference's "3x GF" chain of CLI runs).
 *   workspace: device scratch of at least rf_gf_workspace_bytes(1, ...) bytes; larger
 *   workspaces let more images of the batch be in flight at once.
 *   stream: all work is ordered after what `stream` holds at the call and before what is
 *   enqueued on it afterwards; inside, half of a batch may run on a side stream of the library
 *   that is forked from and joined back into `stream` with events (graph capture keeps working).
 *   Side streams are per caller stream (at most 16 are kept), so concurrent callers on different
 *   streams - eager or capturing - never meet on one.
 */
size_t rf_gf_workspace_bytes(int n, int h, int w, int guide_cn, int src_cn, int radius);
int rf_gf_u8(const uint8_t *guide, const uint8_t *src, uint8_t *dst, int n, int h, int w,
             int guide_cn, int src_cn, int radius, double eps, int iterations, void *workspace,
             size_t workspace_bytes, void *stream);

/*
 * 1x1 CNN reflectance predictor on uint8 BGR images.
 * Replaces  caffe.Net(network_definition.prototxt, TEST, weights=learned_weights.caffemodel),
 * blobs['images'] <- imgCV2_to_caffeBlob(image), forward(), blobs['reflectance_intensity']
 * (/root/reference/decompose_with_trained_CNN.py:57-69, 82-95, 100-106).
 *   bgr        n*h*w*3 uint8 device (cv2.imread layout)
 *   r_out      n*h*w float32 device or NULL: sigmoid output in (0,1)
 *   r_u8_out   n*h*w uint8 device or NULL: trunc(r*255), the bytes of `<base>-r.png`
 *              (/root/reference/image_utils.py:63-68)
 *   weights    4513 float32 on the device:
 *              W0[32][3] b0[32] | 4 x (W[32][32] b[32]) | wf[160] bf[1]
 *   srgb_lut   256 float32 on the device: linear value of each sRGB byte
 */
#define RF_CNN_NPARAMS 4513
#define RF_CNN_NPACKED 4673 /* floats of rf_cnn_pack_weights' output */
int rf_cnn_reflectance_u8(const uint8_t *bgr, float *r_out, uint8_t *r_u8_out, int n, int h, int w,
                          const float *weights, const float *srgb_lut, void *stream);
/*
 * The same forward pass split the way pycaffe splits it: rf_cnn_pack_weights is the net-load step
 * (caffe.Net(prototxt, TEST, weights=...), /root/reference/decompose_with_trained_CNN.py:104-106),
 * done once per set of weights; rf_cnn_reflectance_packed_u8 is forward() on a loaded net
 * (:86-92).  The library keeps no state for this pair, so it can be captured into a HIP graph and
 * used from any number of streams.  rf_cnn_reflectance_u8 above is pack + forward in one call,
 * with the packed copy in a small library-owned table keyed by (device, stream) whose slots are
 * recycled; it is refused (RF_E_UNSUPPORTED) on a stream that is being captured - a graph would
 * keep a slot's address after the slot has moved on - so graphs use the pair.
 *   weights  4513 float32 on the device, layout as above
 *   packed   RF_CNN_NPACKED float32 on the device, caller-owned: the weights in the order the
 *            kernel streams them, the 160 fuse weights twice each (opaque; valid for this library
 *            version)
 */
int rf_cnn_pack_weights(const float *weights, float *packed, void *stream);
int rf_cnn_reflectance_packed_u8(const uint8_t *bgr, float *r_out, uint8_t *r_u8_out, int n, int h,
                                 int w, const float *packed, const float *srgb_lut, void *stream);

/*
 * Colourised reflectance and shading PNG bytes of decompose_image.
 * Replaces the host numpy chain  iu.colorize(reflectance_gray, image)  +  iu.imwrite(..., sRGB=True)
 * of both results (/root/reference/decompose_with_trained_CNN.py:121-128,
 * /root/reference/image_utils.py:42-49, 60-92) for a batch that is already on the device:
 *   shading = mean_c(bgr) / r ; reflectance = bgr / max(shading, 1e-3)          (float64)
 *   each: if max > 1: clip(x / percentile(x, 99.9, 'lower'), 0, 1); rgb_to_srgb; trunc(x * 255)
 *   bgr          n*h*w*3 uint8 device        r   n*h*w float32 device (the CNN output)
 *   refl_out     n*h*w*3 uint8 device or NULL (bytes of `<base>-r_colorized.png`, BGR order)
 *   shading_out  n*h*w   uint8 device or NULL (bytes of `<base>-s_colorized.png`)
 *   k_refl, k_shading  0-based rank of the 99.9-percentile ('lower') among the 3*h*w resp. h*w
 *                values of one image, computed by the caller with numpy's own index rule
 *   srgb_steps   255 float64 on the device: srgb_steps[k-1] = smallest x in (0.0031308, 1] with
 *                trunc(((1.055*x)^(1/2.4) - 0.055) * 255) >= k under the HOST's pow (+inf if none);
 *                this makes the bytes exact for the libm the reference would have used
 *   workspace    device scratch, rf_colorize_workspace_bytes(n) bytes
 */
size_t rf_colorize_workspace_bytes(int n);
int rf_colorize_srgb_u8(const uint8_t *bgr, const float *r, uint8_t *refl_out, uint8_t *shading_out,
                        int n, int h, int w, unsigned long long k_refl, unsigned long long k_shading,
                        const double *srgb_steps, void *workspace, size_t workspace_bytes,
                        void *stream);

/*
 * CV_32F variants of the two filters (never reached by the reference's CLIs, whose images come
 * from cv2.imread as uint8; provided so that callers which stop quantising between stages keep
 * the cv2.ximgproc semantics).  Same layouts as the 8-bit entry points with float pixels.
 *   rf_jbf_f32: jointBilateralFilter_32f - colour weight interpolated in a table of 4096 bins per
 *     joint channel over the joint's value range, built on the host with libm's exp: the range
 *     comes back to the host, so THIS ENTRY POINT SYNCHRONISES THE STREAM.  A constant joint
 *     image returns RF_E_UNSUPPORTED (OpenCV switches to a Gaussian blur there), and so does
 *     BORDER_CONSTANT (its zero padding indexes OpenCV's table out of bounds).
 *   rf_gf_f32: guidedFilter on float guide/src, float result (no rounding), `iterations` chained.
 */
size_t rf_jbf_f32_workspace_bytes(int n, int joint_cn);
int rf_jbf_f32(const float *joint, const float *src, float *dst, int n, int h, int w, int joint_cn,
               int src_cn, int d, double sigma_color, double sigma_space, int border,
               void *workspace, size_t workspace_bytes, void *stream);
size_t rf_gf_f32_workspace_bytes(int n, int h, int w, int guide_cn, int src_cn, int radius);
int rf_gf_f32(const float *guide, const float *src, float *dst, int n, int h, int w, int guide_cn,
              int src_cn, int radius, double eps, int iterations, void *workspace,
              size_t workspace_bytes, void *stream);

/*
 * WHDR (weighted human disagreement rate, Bell et al. 2014) of a batch of reflectance predictions.
 * Replaces the per-comparison Python loop  whdr(reflectance, comparisons, delta)  of
 * /root/reference/training/layers/whdr_layer.py:253-287 (lightness: :180-196).
 *   refl     n*c*h*w float32 device, planar [n][c][h][w], c in {1,3}
 *   points   total*5 int32 device: x1, y1, x2, y2 (pixels, inside the image), darker (0 'E', 1, 2)
 *   weights  total float64 device: darker_score of each comparison
 *   offsets  n+1 int32 device: comparisons of image i are [offsets[i], offsets[i+1])
 *   out      n float64 device: error_sum / weight_sum (0 for an image without comparisons)
 * Lightness and ratios are float32 and compared with (float)(1 + delta); the two sums are float64,
 * accumulated in comparison order.
 */
int rf_whdr_f32(const float *refl, int n, int c, int h, int w, const int *points,
                const double *weights, const int *offsets, double delta, double *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* REFLECTANCE_FILTERING_H */
