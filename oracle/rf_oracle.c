/*
 * rf_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the arithmetic behind the reference's hot path:
 *   /root/reference/filter_reflectance.py:60-64   cv2.ximgproc.jointBilateralFilter
 *   /root/reference/filter_reflectance.py:67-70   cv2.ximgproc.guidedFilter
 *   /root/reference/decompose_with_trained_CNN.py:82-95  caffe Net.forward()
 *   /root/reference/decompose_with_trained_CNN.py:57-69  imgCV2_to_caffeBlob
 *   /root/reference/image_utils.py:60-68          imwrite's truncating uint8 cast
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (reflectance_filtering_amd) never does.
 *
 * PARITY UNPINNED for the two filters and the CNN forward: the arithmetic of
 * those three calls lives in third-party code that is neither vendored in the
 * reference nor installed here (OpenCV + opencv_contrib "ximgproc", version hint
 * 3.1.0 at filter_reflectance.py:37-43; BVLC Caffe, decompose_with_trained_CNN.py:41-46).
 * The reference holds no tests or golden vectors.  What is
 * restated below is the published algorithm of
 *   opencv_contrib/modules/ximgproc/src/joint_bilateral_filter.cpp  (8u path)
 *   opencv_contrib/modules/ximgproc/src/guided_filter.cpp           (3-ch guide)
 *   opencv_contrib/modules/ximgproc/src/edgeaware_filters_common.cpp (mul/sub_mul/...)
 *   opencv/modules/imgproc/src/smooth.cpp  (boxFilter: RowSum<float,double>,
 *                                           ColumnSum<double,float>)
 *   opencv/modules/core  borderInterpolate, cvRound, saturate_cast, Matx operator/
 *   caffe/src/caffe/layers/{base_conv,relu,concat,sigmoid}_layer.cpp
 * in their (non-FMA, x86-64 baseline) operation order.  It is pinned against
 * (a) the float64 definitions in oracle/t0_numpy.py, (b) known-answer cases and
 * (c) the reference's own Python helpers for every step the reference does
 * itself (tests/golden/).
 *
 * The recalled choices a real OpenCV / Caffe could overturn each have a switch (RFO_VAR_*,
 * rfo_set_variants): tools/t2_report.py tries the combinations and names the matching one.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 * (-ffp-contract=off matters: every mul/add below is a separately rounded
 * IEEE-754 binary32/binary64 operation, as in an SSE2 build of OpenCV).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RFO_BORDER_CONSTANT 0
#define RFO_BORDER_REPLICATE 1
#define RFO_BORDER_REFLECT 2
#define RFO_BORDER_WRAP 3
#define RFO_BORDER_REFLECT_101 4

#define RFO_FLAG_TRUE_DIVISION 1 /* dst = sum / wsum instead of sum * (1.f / wsum) */

/*
 * VARIANTS.  The choices below were RECALLED from the upstream sources, not read: a real OpenCV /
 * Caffe build may have made the other one (another version, an FMA-contracting compiler, another
 * dispatch path).  Each has a switch, so that tools/t2_report.py can try the combinations against
 * a real cv2 / caffe and name the one that matches bit for bit; the default (0) is the restatement
 * the HIP kernels and the frozen vectors follow.  Process-global; set by rfo_set_variants().
 */
#define RFO_VAR_JBF_TRUE_DIVISION 0x01 /* joint bilateral: dst = sum / wsum (not sum * (1.f / wsum)) */
#define RFO_VAR_JBF_FMA 0x02           /* joint bilateral: sum = fma(w, src, sum) (an FMA build) */
#define RFO_VAR_GF_DIAG_ADD_EPS 0x04   /* guided: cov_ii = (mean(I*I) - mean*mean) + eps, not sub_mad(-eps) */
#define RFO_VAR_GF_FLOAT_BOXSUM 0x08   /* guided: boxFilter running sums in float, not double */
#define RFO_VAR_GF_FMA 0x10            /* guided: the mul/sub_mul/add_mul/sub_mad helpers contracted to FMAs */
#define RFO_VAR_CNN_SIGMOID_TANH 0x20  /* CNN: sigmoid = 0.5 * tanh(0.5 x) + 0.5 (later BVLC master) */
#define RFO_VAR_CNN_GEMM_NO_FMA 0x40   /* CNN: dot products as separately rounded mul and add */
#define RFO_VAR_ALL 0x7f
static unsigned g_rfo_variants = 0;
unsigned rfo_set_variants(unsigned mask)
{
    unsigned old = g_rfo_variants;
    g_rfo_variants = mask & RFO_VAR_ALL;
    return old;
}
unsigned rfo_get_variants(void) { return g_rfo_variants; }

int rfo_version(void) { return 1; }

/* cv::borderInterpolate (opencv/modules/core/src/copy.cpp).  Returns -1 for
 * BORDER_CONSTANT outside the image. */
int rfo_border_interpolate(int p, int len, int border)
{
    if ((unsigned)p < (unsigned)len)
        return p;
    if (border == RFO_BORDER_REPLICATE)
        return p < 0 ? 0 : len - 1;
    if (border == RFO_BORDER_REFLECT || border == RFO_BORDER_REFLECT_101) {
        int delta = border == RFO_BORDER_REFLECT_101;
        if (len == 1)
            return 0;
        do {
            if (p < 0)
                p = -p - 1 + delta;
            else
                p = len - 1 - (p - len) - delta;
        } while ((unsigned)p >= (unsigned)len);
        return p;
    }
    if (border == RFO_BORDER_WRAP) {
        if (p < 0)
            p -= ((p - len + 1) / len) * len;
        if (p >= len)
            p %= len;
        return p;
    }
    return -1;
}

/* cvRound(double): round-half-to-even (SSE2 cvtsd2si under the default mode). */
static inline int rfo_cv_round(double v) { return (int)lrint(v); }
static inline int rfo_cv_roundf(float v) { return (int)lrintf(v); }

static inline uint8_t rfo_saturate_u8(float v)
{
    int iv = rfo_cv_roundf(v);
    return (uint8_t)(iv < 0 ? 0 : (iv > 255 ? 255 : iv));
}

int rfo_jbf_radius(int d, double sigma_space)
{
    int radius;
    if (sigma_space <= 0)
        sigma_space = 1;
    radius = d <= 0 ? rfo_cv_round(sigma_space * 1.5) : d / 2;
    return radius < 1 ? 1 : radius;
}

/* Tap table of jointBilateralFilter_8u: row-major over the (2r+1)^2 window,
 * keeping taps with sqrt(i*i+j*j) <= radius.  Returns maxk; di/dj/sw may be NULL. */
int rfo_jbf_taps(int radius, double sigma_space, int *di, int *dj, float *sw)
{
    double gauss_space_coeff;
    int maxk = 0;
    if (sigma_space <= 0)
        sigma_space = 1;
    gauss_space_coeff = -0.5 / (sigma_space * sigma_space);
    for (int i = -radius; i <= radius; i++)
        for (int j = -radius; j <= radius; j++) {
            double r = sqrt((double)i * i + (double)j * j);
            if (r > radius)
                continue;
            if (sw)
                sw[maxk] = (float)exp(r * r * gauss_space_coeff);
            if (di)
                di[maxk] = i;
            if (dj)
                dj[maxk] = j;
            maxk++;
        }
    return maxk;
}

void rfo_jbf_color_lut(double sigma_color, int joint_cn, float *lut /* 256*joint_cn */)
{
    double gauss_color_coeff;
    if (sigma_color <= 0)
        sigma_color = 1;
    gauss_color_coeff = -0.5 / (sigma_color * sigma_color);
    for (int i = 0; i < 256 * joint_cn; i++)
        lut[i] = (float)exp(i * i * gauss_color_coeff);
}

/*
 * jointBilateralFilter, CV_8U joint and src (joint_bilateral_filter.cpp,
 * jointBilateralFilter_8u + JointBilateralFilter_8u::operator()).
 * joint: h*w*joint_cn, src/dst: h*w*src_cn, all interleaved, tightly packed.
 * Returns 0, or -1 on bad arguments.
 */
int rfo_jbf_u8(const uint8_t *joint, const uint8_t *src, uint8_t *dst, int h, int w, int joint_cn,
               int src_cn, int d, double sigma_color, double sigma_space, int border, int flags,
               int threads)
{
    if (!joint || !src || !dst || h <= 0 || w <= 0)
        return -1;
    if ((joint_cn != 1 && joint_cn != 3) || (src_cn != 1 && src_cn != 3))
        return -1;
    if (border < 0 || border > 4)
        return -1;
    int radius = rfo_jbf_radius(d, sigma_space);
    int dd = 2 * radius + 1;
    int *di = (int *)malloc(sizeof(int) * dd * dd);
    int *dj = (int *)malloc(sizeof(int) * dd * dd);
    float *sw = (float *)malloc(sizeof(float) * dd * dd);
    float *lut = (float *)malloc(sizeof(float) * 256 * joint_cn);
    int maxk = rfo_jbf_taps(radius, sigma_space, di, dj, sw);
    rfo_jbf_color_lut(sigma_color, joint_cn, lut);

    /* copyMakeBorder(joint/src, radius on all sides, borderType): padded copies. */
    int ph = h + 2 * radius, pw = w + 2 * radius;
    uint8_t *jp = (uint8_t *)malloc((size_t)ph * pw * joint_cn);
    uint8_t *sp = (uint8_t *)malloc((size_t)ph * pw * src_cn);
    for (int y = 0; y < ph; y++) {
        int sy = rfo_border_interpolate(y - radius, h, border);
        for (int x = 0; x < pw; x++) {
            int sx = rfo_border_interpolate(x - radius, w, border);
            for (int c = 0; c < joint_cn; c++)
                jp[((size_t)y * pw + x) * joint_cn + c] =
                    (sy < 0 || sx < 0) ? 0 : joint[((size_t)sy * w + sx) * joint_cn + c];
            for (int c = 0; c < src_cn; c++)
                sp[((size_t)y * pw + x) * src_cn + c] =
                    (sy < 0 || sx < 0) ? 0 : src[((size_t)sy * w + sx) * src_cn + c];
        }
    }
    int *ofs = (int *)malloc(sizeof(int) * maxk);
    for (int k = 0; k < maxk; k++)
        ofs[k] = di[k] * pw + dj[k];
    const int var_fma = (g_rfo_variants & RFO_VAR_JBF_FMA) != 0;
    if (g_rfo_variants & RFO_VAR_JBF_TRUE_DIVISION)
        flags |= RFO_FLAG_TRUE_DIVISION;

#ifdef _OPENMP
    if (threads <= 0)
        threads = omp_get_max_threads();
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4)
#endif
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            size_t c0 = (size_t)(y + radius) * pw + (x + radius);
            const uint8_t *jc = jp + c0 * joint_cn;
            int j0[3] = {0, 0, 0};
            for (int c = 0; c < joint_cn; c++)
                j0[c] = jc[c];
            float sum[3] = {0.0f, 0.0f, 0.0f};
            float wsum = 0.0f;
            for (int k = 0; k < maxk; k++) {
                const uint8_t *jt = jp + (c0 + ofs[k]) * joint_cn;
                const uint8_t *st = sp + (c0 + ofs[k]) * src_cn;
                int alpha = 0;
                for (int c = 0; c < joint_cn; c++)
                    alpha += abs(j0[c] - (int)jt[c]);
                float weight = sw[k] * lut[alpha];
                if (var_fma) {
                    for (int c = 0; c < src_cn; c++)
                        sum[c] = fmaf(weight, (float)st[c], sum[c]);
                } else {
                    for (int c = 0; c < src_cn; c++) {
                        float prod = weight * (float)st[c]; /* separately rounded mul ... */
                        sum[c] = sum[c] + prod;              /* ... then add (no FMA)      */
                    }
                }
                wsum = wsum + weight;
            }
            uint8_t *o = dst + ((size_t)y * w + x) * src_cn;
            if (flags & RFO_FLAG_TRUE_DIVISION) {
                for (int c = 0; c < src_cn; c++)
                    o[c] = rfo_saturate_u8(sum[c] / wsum);
            } else {
                /* Vec<float,cn> / float  ==  a * (1.f / alpha)   (core/matx.hpp) */
                float inv = 1.0f / wsum;
                for (int c = 0; c < src_cn; c++)
                    o[c] = rfo_saturate_u8(sum[c] * inv);
            }
        }
    }
    free(ofs);
    free(jp);
    free(sp);
    free(di);
    free(dj);
    free(sw);
    free(lut);
    return 0;
}

/*
 * jointBilateralFilter, CV_32F joint and src (jointBilateralFilter_32f +
 * JointBilateralFilter_32f::operator(), the structure of imgproc's bilateralFilter_32f): the colour
 * weight is linearly interpolated in a table of 4096 bins per joint channel spanning the joint's
 * value range.  RECALLED from the OpenCV sources, like the rest of this file.
 * Returns 0, -1 on bad arguments, -2 when the joint is constant (OpenCV then falls back to a
 * Gaussian blur, which is not restated here).
 */
int rfo_jbf_f32(const float *joint, const float *src, float *dst, int h, int w, int joint_cn,
                int src_cn, int d, double sigma_color, double sigma_space, int border, int threads)
{
    if (!joint || !src || !dst || h <= 0 || w <= 0)
        return -1;
    if ((joint_cn != 1 && joint_cn != 3) || (src_cn != 1 && src_cn != 3))
        return -1;
    /* BORDER_CONSTANT pads with 0, outside the joint's value range: OpenCV's table index then
     * runs out of bounds (undefined behaviour), so that border type is not restated */
    if (border < 1 || border > 4)
        return -1;
    if (sigma_color <= 0)
        sigma_color = 1;
    int radius = rfo_jbf_radius(d, sigma_space);
    int dd = 2 * radius + 1;
    size_t npx = (size_t)h * w;
    double minv = joint[0], maxv = joint[0];
    for (size_t i = 0; i < npx * joint_cn; i++) {
        if (joint[i] < minv)
            minv = joint[i];
        if (joint[i] > maxv)
            maxv = joint[i];
    }
    if (fabs(minv - maxv) < FLT_EPSILON)
        return -2;
    const int bins = (1 << 12) * joint_cn;
    float len = (float)(maxv - minv) * joint_cn;
    float scale_index = bins / len;
    double gauss_color_coeff = -0.5 / (sigma_color * sigma_color);
    float *lut = (float *)malloc(sizeof(float) * (bins + 2));
    float last = 1.f;
    for (int i = 0; i < bins + 2; i++) {
        if (last > 0.f) {
            double val = i / scale_index;
            lut[i] = (float)exp(val * val * gauss_color_coeff);
            last = lut[i];
        } else {
            lut[i] = 0.f;
        }
    }
    int *di = (int *)malloc(sizeof(int) * dd * dd);
    int *dj = (int *)malloc(sizeof(int) * dd * dd);
    float *sw = (float *)malloc(sizeof(float) * dd * dd);
    int maxk = rfo_jbf_taps(radius, sigma_space, di, dj, sw);
    int ph = h + 2 * radius, pw = w + 2 * radius;
    float *jp = (float *)malloc(sizeof(float) * (size_t)ph * pw * joint_cn);
    float *sp = (float *)malloc(sizeof(float) * (size_t)ph * pw * src_cn);
    for (int y = 0; y < ph; y++) {
        int sy = rfo_border_interpolate(y - radius, h, border);
        for (int x = 0; x < pw; x++) {
            int sx = rfo_border_interpolate(x - radius, w, border);
            for (int c = 0; c < joint_cn; c++)
                jp[((size_t)y * pw + x) * joint_cn + c] =
                    (sy < 0 || sx < 0) ? 0.f : joint[((size_t)sy * w + sx) * joint_cn + c];
            for (int c = 0; c < src_cn; c++)
                sp[((size_t)y * pw + x) * src_cn + c] =
                    (sy < 0 || sx < 0) ? 0.f : src[((size_t)sy * w + sx) * src_cn + c];
        }
    }
    int *ofs = (int *)malloc(sizeof(int) * maxk);
    for (int k = 0; k < maxk; k++)
        ofs[k] = di[k] * pw + dj[k];
#ifdef _OPENMP
    if (threads <= 0)
        threads = omp_get_max_threads();
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4)
#endif
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            size_t c0 = (size_t)(y + radius) * pw + (x + radius);
            const float *jc = jp + c0 * joint_cn;
            float sum[3] = {0.0f, 0.0f, 0.0f};
            float wsum = 0.0f;
            for (int k = 0; k < maxk; k++) {
                const float *jt = jp + (c0 + ofs[k]) * joint_cn;
                const float *st = sp + (c0 + ofs[k]) * src_cn;
                float alpha = 0.0f;
                for (int c = 0; c < joint_cn; c++)
                    alpha = alpha + fabsf(jc[c] - jt[c]);
                alpha = alpha * scale_index;
                int idx = (int)alpha;
                alpha = alpha - (float)idx;
                float diff = lut[idx + 1] - lut[idx];
                float interp = alpha * diff;
                interp = lut[idx] + interp;
                float weight = sw[k] * interp;
                for (int c = 0; c < src_cn; c++) {
                    float prod = weight * st[c];
                    sum[c] = sum[c] + prod;
                }
                wsum = wsum + weight;
            }
            float inv = 1.0f / wsum;
            for (int c = 0; c < src_cn; c++)
                dst[((size_t)y * w + x) * src_cn + c] = sum[c] * inv;
        }
    }
    free(ofs);
    free(jp);
    free(sp);
    free(di);
    free(dj);
    free(sw);
    free(lut);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Guided filter                                                             */
/* ------------------------------------------------------------------------- */

/*
 * cv::boxFilter(src, CV_32F, Size(2r+1,2r+1), anchor centre, normalize=true,
 * BORDER_REFLECT) on one CV_32FC1 plane.  For 32F input the sum type is 64F:
 * RowSum<float,double> is a running sum along the border-extended row,
 * ColumnSum<double,float> a running sum of row sums down the image that starts
 * ksize-1 rows above the first output row; each output is (float)(s * scale).
 * src and dst may alias.
 */
static void rfo_box_mean_float_sums(const float *src, float *dst, int h, int w, int r);
static void rfo_box_mean(const float *src, float *dst, int h, int w, int r)
{
    if (g_rfo_variants & RFO_VAR_GF_FLOAT_BOXSUM) {
        rfo_box_mean_float_sums(src, dst, h, w, r);
        return;
    }
    int ks = 2 * r + 1;
    double scale = 1.0 / ((double)ks * (double)ks);
    double *rows = (double *)malloc(sizeof(double) * (size_t)h * w);
    int ew = w + ks - 1;
    float *ext = (float *)malloc(sizeof(float) * ew);
    int *xtab = (int *)malloc(sizeof(int) * ew);
    for (int x = 0; x < ew; x++)
        xtab[x] = rfo_border_interpolate(x - r, w, RFO_BORDER_REFLECT);
    for (int y = 0; y < h; y++) {
        const float *S0 = src + (size_t)y * w;
        double *D = rows + (size_t)y * w;
        for (int x = 0; x < ew; x++)
            ext[x] = S0[xtab[x]];
        double s = 0;
        for (int i = 0; i < ks; i++)
            s += (double)ext[i];
        D[0] = s;
        for (int i = 0; i < w - 1; i++) {
            s += (double)ext[i + ks] - (double)ext[i];
            D[i + 1] = s;
        }
    }
    double *SUM = (double *)calloc(w, sizeof(double));
    /* first ksize-1 extended rows: source rows -r .. r-1 */
    for (int yy = -r; yy < r; yy++) {
        const double *Sp = rows + (size_t)rfo_border_interpolate(yy, h, RFO_BORDER_REFLECT) * w;
        for (int i = 0; i < w; i++)
            SUM[i] += Sp[i];
    }
    for (int y = 0; y < h; y++) {
        const double *Sp = rows + (size_t)rfo_border_interpolate(y + r, h, RFO_BORDER_REFLECT) * w;
        const double *Sm = rows + (size_t)rfo_border_interpolate(y - r, h, RFO_BORDER_REFLECT) * w;
        float *D = dst + (size_t)y * w;
        for (int i = 0; i < w; i++) {
            double s0 = SUM[i] + Sp[i];
            D[i] = (float)(s0 * scale);
            SUM[i] = s0 - Sm[i];
        }
    }
    free(SUM);
    free(xtab);
    free(ext);
    free(rows);
}

/*
 * Exactness census of the stage-2 box means (tools/gf_exactness.py; measurement aid, not part of
 * the restatement).  rfo_box_mean's arithmetic with every double addition / subtraction checked by
 * TwoSum: an operation is "inexact" when it rounded.  A chain none of whose operations rounded
 * holds the mathematically exact window sum, i.e. a value that does not depend on the order.
 *   c[0] rows              c[1] rows with a rounded operation
 *   c[2] row operations    c[3] rounded row operations
 *   c[4] columns           c[5] columns with a rounded operation
 *   c[6] column operations c[7] rounded column operations
 *   c[8] rows that pass the order-free sufficient test: every non-zero value of the row is a
 *        multiple of 2^e (e = its float exponent - 23) and ks * max|v| < 2^(e_min + 53)
 *   c[9] planes            c[10] planes with no rounded operation at all
 *   c[11] blocks of 64 rows  c[12] blocks that pass the JOINT test the GPU's stage 1 evaluates per
 *        (64-row block, plane): exponent fields of the largest and of the smallest non-zero
 *        magnitude of the whole block, Emax - max(Emin, 1) <= 29 - ceil(log2 ks)
 *        (reflectance_filtering_amd/csrc/rf_gf.hip, "exact rows")
 */
static unsigned long long g_rfo_census[16];
static int g_rfo_census_on;
void rfo_census(int on, unsigned long long *out16)
{
    if (out16)
        for (int i = 0; i < 16; i++)
            out16[i] = g_rfo_census[i];
    if (on)
        for (int i = 0; i < 16; i++)
            g_rfo_census[i] = 0;
    g_rfo_census_on = on;
}
static inline int rfo_add_rounded(double a, double b, double s)
{
    double bb = s - a;
    double err = (a - (s - bb)) + (b - bb);
    return err != 0.0;
}
static void rfo_box_mean_census(const float *src, float *dst, int h, int w, int r)
{
    unsigned long long c[16] = {0};
    int ks = 2 * r + 1;
    double scale = 1.0 / ((double)ks * (double)ks);
    double *rows = (double *)malloc(sizeof(double) * (size_t)h * w);
    int ew = w + ks - 1;
    float *ext = (float *)malloc(sizeof(float) * ew);
    int lim = 29;
    while ((1 << (29 - lim)) < ks)
        lim--;
    unsigned bmax = 0, bmin = 0xffffffffu;
    for (int y = 0; y < h; y++) {
        const float *S0 = src + (size_t)y * w;
        double *D = rows + (size_t)y * w;
        int emin = 10000;
        float vmax = 0.f;
        for (int x = 0; x < w; x++) {
            unsigned u;
            memcpy(&u, &S0[x], 4);
            u &= 0x7fffffffu;
            if (u > bmax)
                bmax = u;
            if (u != 0 && u < bmin)
                bmin = u;
        }
        if ((y & 63) == 63 || y == h - 1) {
            int e1 = (int)(bmax >> 23), e0 = (int)(bmin >> 23);
            if (e0 < 1)
                e0 = 1;
            if (e1 < 1)
                e1 = 1;
            c[11]++;
            c[12] += (bmax == 0 || e1 - e0 <= lim);
            bmax = 0;
            bmin = 0xffffffffu;
        }
        for (int x = 0; x < ew; x++) {
            float v = S0[rfo_border_interpolate(x - r, w, RFO_BORDER_REFLECT)];
            ext[x] = v;
            if (v != 0.f) {
                int e;
                (void)frexpf(v, &e); /* |v| in [2^(e-1), 2^e): lsb weight 2^(e-24) */
                if (e - 24 < emin)
                    emin = e - 24;
                if (fabsf(v) > vmax)
                    vmax = fabsf(v);
            }
        }
        if (vmax == 0.f || (double)ks * (double)vmax < ldexp(1.0, emin + 53))
            c[8]++;
        unsigned long long bad = 0, ops = 0;
        double s = 0;
        for (int i = 0; i < ks; i++) {
            double t = s + (double)ext[i];
            bad += rfo_add_rounded(s, (double)ext[i], t);
            ops++;
            s = t;
        }
        D[0] = s;
        for (int i = 0; i < w - 1; i++) {
            double a = (double)ext[i + ks], b = (double)ext[i];
            double d = a - b;
            bad += rfo_add_rounded(a, -b, d);
            double t = s + d;
            bad += rfo_add_rounded(s, d, t);
            ops += 2;
            s = t;
            D[i + 1] = s;
        }
        c[0]++;
        c[1] += bad != 0;
        c[2] += ops;
        c[3] += bad;
    }
    double *SUM = (double *)calloc(w, sizeof(double));
    unsigned long long *cbad = (unsigned long long *)calloc(w, sizeof(unsigned long long));
    for (int yy = -r; yy < r; yy++) {
        const double *Sp = rows + (size_t)rfo_border_interpolate(yy, h, RFO_BORDER_REFLECT) * w;
        for (int i = 0; i < w; i++) {
            double t = SUM[i] + Sp[i];
            cbad[i] += rfo_add_rounded(SUM[i], Sp[i], t);
            SUM[i] = t;
        }
        c[6] += w;
    }
    for (int y = 0; y < h; y++) {
        const double *Sp = rows + (size_t)rfo_border_interpolate(y + r, h, RFO_BORDER_REFLECT) * w;
        const double *Sm = rows + (size_t)rfo_border_interpolate(y - r, h, RFO_BORDER_REFLECT) * w;
        float *D = dst + (size_t)y * w;
        for (int i = 0; i < w; i++) {
            double s0 = SUM[i] + Sp[i];
            cbad[i] += rfo_add_rounded(SUM[i], Sp[i], s0);
            D[i] = (float)(s0 * scale);
            double t = s0 - Sm[i];
            cbad[i] += rfo_add_rounded(s0, -Sm[i], t);
            SUM[i] = t;
        }
        c[6] += 2 * (unsigned long long)w;
    }
    for (int i = 0; i < w; i++) {
        c[4]++;
        c[5] += cbad[i] != 0;
        c[7] += cbad[i];
    }
    c[9] = 1;
    c[10] = (c[3] == 0 && c[7] == 0);
    free(cbad);
    free(SUM);
    free(ext);
    free(rows);
#ifdef _OPENMP
#pragma omp critical(rfo_census)
#endif
    for (int i = 0; i < 16; i++)
        g_rfo_census[i] += c[i];
}

/*
 * Exact rows (test infrastructure for the GPU's debug option "gf_exact",
 * reflectance_filtering_amd/csrc/rf_gf_fused.hpp).  For `rows` independent rows of w floats
 * (w a multiple of 16, w >= 16), per row:
 *   pass[y]    the sufficient test as the GPU evaluates it on one plane: mx16 = bits of the largest
 *              magnitude >> 15, mn16 = (smallest non-zero magnitude's (bits << 1) - 2) >> 16; passes
 *              if mx16 == 0 or max(mx16 >> 8, 1) - max(mn16 >> 8, 1) <= 29 - ceil(log2 ks) (never with
 *              an infinity or a NaN in the row: exponent field 255)
 *   rounded[y] operations of RowSum<float,double>'s chain (rfo_box_mean's row pass) that rounded
 *   equal[y]   1 if the row sum at every column 16 b, rebuilt the GPU's way - the sums of the aligned
 *              16-column blocks as a xor-butterfly over their 16 values, then prefix of block b + q
 *              (its sum minus the values behind column 16 b + r), blocks b + q - 1 .. b - q, suffix of
 *              block b - q - 1, block indices reflected like columns - is the chain's double
 * The claim under test: pass => rounded == 0 and equal == 1.
 */
void rfo_exact_rows_check(const float *src, int rows, int w, int r, int *pass, int *rounded, int *equal)
{
    int ks = 2 * r + 1, ew = w + ks - 1, nb = w / 16, q = r / 16, c0 = r % 16;
    int lim = 29;
    while ((1 << (29 - lim)) < ks)
        lim--;
    float *ext = (float *)malloc(sizeof(float) * ew);
    double *D = (double *)malloc(sizeof(double) * w);
    double *F = (double *)malloc(sizeof(double) * nb);
    for (int y = 0; y < rows; y++) {
        const float *S0 = src + (size_t)y * w;
        unsigned bmax = 0, tmin = 0xffffffffu;
        for (int x = 0; x < w; x++) {
            unsigned u;
            memcpy(&u, &S0[x], 4);
            unsigned mag = u & 0x7fffffffu, t = (u << 1) - 2u;
            if (mag > bmax)
                bmax = mag;
            if (t < tmin)
                tmin = t;
        }
        {
            int mx16 = (int)(bmax >> 15), mn16 = (int)(tmin >> 16);
            int e1 = mx16 >> 8, e0 = mn16 >> 8;
            if (e1 < 1)
                e1 = 1;
            if (e0 < 1)
                e0 = 1;
            pass[y] = (mx16 == 0) || (e1 != 255 && e1 - e0 <= lim); /* 255: an infinity or a NaN */
        }
        for (int x = 0; x < ew; x++)
            ext[x] = S0[rfo_border_interpolate(x - r, w, RFO_BORDER_REFLECT)];
        int bad = 0;
        double s = 0;
        for (int i = 0; i < ks; i++) {
            double t = s + (double)ext[i];
            bad += rfo_add_rounded(s, (double)ext[i], t);
            s = t;
        }
        D[0] = s;
        for (int i = 0; i < w - 1; i++) {
            double a = (double)ext[i + ks], b = (double)ext[i];
            double d = a - b;
            bad += rfo_add_rounded(a, -b, d);
            double t = s + d;
            bad += rfo_add_rounded(s, d, t);
            s = t;
            D[i + 1] = s;
        }
        rounded[y] = bad;
        for (int k = 0; k < nb; k++) {
            double v[16];
            for (int i = 0; i < 16; i++)
                v[i] = (double)S0[16 * k + i];
            for (int m = 1; m < 16; m <<= 1) {
                double n[16];
                for (int i = 0; i < 16; i++)
                    n[i] = v[i] + v[i ^ m];
                memcpy(v, n, sizeof(v));
            }
            F[k] = v[0];
        }
        int eq = 1;
        for (int b = 0; b < nb; b++) {
            double se = F[rfo_border_interpolate(b + q, nb, RFO_BORDER_REFLECT)];
            for (int c = 1; c <= 15 - c0; c++)
                se -= (double)S0[rfo_border_interpolate(16 * b + c + r, w, RFO_BORDER_REFLECT)];
            for (int k = b + q - 1; k >= b - q; k--)
                se += F[rfo_border_interpolate(k, nb, RFO_BORDER_REFLECT)];
            for (int c = 1; c <= c0; c++)
                se += (double)S0[rfo_border_interpolate(16 * b + c - 1 - r, w, RFO_BORDER_REFLECT)];
            if (memcmp(&se, &D[16 * b], sizeof(double)) != 0)
                eq = 0;
        }
        equal[y] = eq;
    }
    free(F);
    free(D);
    free(ext);
}

/* RFO_VAR_GF_FLOAT_BOXSUM: the same two running sums held in float (RowSum<float,float>,
 * ColumnSum<float,float>: out = s * (float)scale) */
static void rfo_box_mean_float_sums(const float *src, float *dst, int h, int w, int r)
{
    int ks = 2 * r + 1;
    float scale = (float)(1.0 / ((double)ks * (double)ks));
    float *rows = (float *)malloc(sizeof(float) * (size_t)h * w);
    int ew = w + ks - 1;
    float *ext = (float *)malloc(sizeof(float) * ew);
    for (int y = 0; y < h; y++) {
        const float *S0 = src + (size_t)y * w;
        float *D = rows + (size_t)y * w;
        for (int x = 0; x < ew; x++)
            ext[x] = S0[rfo_border_interpolate(x - r, w, RFO_BORDER_REFLECT)];
        float s = 0;
        for (int i = 0; i < ks; i++)
            s += ext[i];
        D[0] = s;
        for (int i = 0; i < w - 1; i++) {
            s += ext[i + ks] - ext[i];
            D[i + 1] = s;
        }
    }
    float *SUM = (float *)calloc(w, sizeof(float));
    for (int yy = -r; yy < r; yy++) {
        const float *Sp = rows + (size_t)rfo_border_interpolate(yy, h, RFO_BORDER_REFLECT) * w;
        for (int i = 0; i < w; i++)
            SUM[i] += Sp[i];
    }
    for (int y = 0; y < h; y++) {
        const float *Sp = rows + (size_t)rfo_border_interpolate(y + r, h, RFO_BORDER_REFLECT) * w;
        const float *Sm = rows + (size_t)rfo_border_interpolate(y - r, h, RFO_BORDER_REFLECT) * w;
        float *D = dst + (size_t)y * w;
        for (int i = 0; i < w; i++) {
            float s0 = SUM[i] + Sp[i];
            D[i] = s0 * scale;
            SUM[i] = s0 - Sm[i];
        }
    }
    free(SUM);
    free(ext);
    free(rows);
}

/* edgeaware_filters_common.cpp element-wise helpers (RFO_VAR_GF_FMA: contracted forms) */
static void ew_mul(float *d, const float *a, const float *b, size_t n)
{
    for (size_t i = 0; i < n; i++)
        d[i] = a[i] * b[i];
}
static void ew_sub_mul(float *d, const float *a, const float *b, size_t n)
{
    if (g_rfo_variants & RFO_VAR_GF_FMA) {
        for (size_t i = 0; i < n; i++)
            d[i] = fmaf(-a[i], b[i], d[i]);
        return;
    }
    for (size_t i = 0; i < n; i++) {
        float p = a[i] * b[i];
        d[i] = d[i] - p;
    }
}
static void ew_add_mul(float *d, const float *a, const float *b, size_t n)
{
    if (g_rfo_variants & RFO_VAR_GF_FMA) {
        for (size_t i = 0; i < n; i++)
            d[i] = fmaf(a[i], b[i], d[i]);
        return;
    }
    for (size_t i = 0; i < n; i++) {
        float p = a[i] * b[i];
        d[i] = d[i] + p;
    }
}
static void ew_sub_mad(float *d, const float *a, const float *b, float c0, size_t n)
{
    if (g_rfo_variants & RFO_VAR_GF_FMA) {
        for (size_t i = 0; i < n; i++)
            d[i] = d[i] - fmaf(a[i], b[i], c0);
        return;
    }
    for (size_t i = 0; i < n; i++) {
        float p = a[i] * b[i];
        float q = p + c0;
        d[i] = d[i] - q;
    }
}

/* index of the symmetric pair (i,j) in a 6-entry upper-triangular store */
static inline int sym_idx(int i, int j)
{
    if (i > j) {
        int t = i;
        i = j;
        j = t;
    }
    return i * 3 - i * (i - 1) / 2 + (j - i); /* (0,0)=0 (0,1)=1 (0,2)=2 (1,1)=3 (1,2)=4 (2,2)=5 */
}

/*
 * guidedFilter(guide, src, dst, radius, eps) with CV_8UC3 guide and CV_8UC{1,3}
 * src, dDepth = -1 (uint8 result).  q_f32 (optional) receives the float result
 * before the final convertTo (h*w*src_cn interleaved).
 * Follows GuidedFilterImpl::init / ::filter (guided_filter.cpp).
 */
/* The filter on float data: guide_f h*w*3, src_f h*w*src_cn (interleaved); writes q_f32 and/or the
 * uint8 conversion.  Both depths run exactly this (convertTo(CV_32F) of uint8 data is exact). */
static int gf_core(const float *guide_f, const float *src_f, uint8_t *dst, float *q_f32, int h,
                   int w, int src_cn, int radius, double eps, int threads)
{
    size_t n = (size_t)h * w;
#ifdef _OPENMP
    if (threads <= 0)
        threads = omp_get_max_threads();
#else
    threads = 1;
#endif
    float *I[3], *mI[3], *cov[6], *inv[6];
    for (int c = 0; c < 3; c++) {
        I[c] = (float *)malloc(sizeof(float) * n);
        mI[c] = (float *)malloc(sizeof(float) * n);
        for (size_t i = 0; i < n; i++)
            I[c][i] = guide_f[i * 3 + c]; /* split (+ convertTo(CV_32F), no scaling) */
    }
    for (int k = 0; k < 6; k++) {
        cov[k] = (float *)malloc(sizeof(float) * n);
        inv[k] = (float *)malloc(sizeof(float) * n);
    }
    /* ---- init(guide) ---- */
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (int t = 0; t < 9; t++) {
        if (t < 3) {
            rfo_box_mean(I[t], mI[t], h, w, radius);
        } else {
            static const int P[6][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}};
            int c1 = P[t - 3][0], c2 = P[t - 3][1];
            float *cv = cov[sym_idx(c1, c2)];
            ew_mul(cv, I[c1], I[c2], n);
            rfo_box_mean(cv, cv, h, w, radius);
        }
    }
    float diag = (float)eps;
    for (int c1 = 0; c1 < 3; c1++)
        for (int c2 = c1; c2 < 3; c2++) {
            float *cv = cov[sym_idx(c1, c2)];
            if (c1 != c2) {
                ew_sub_mul(cv, mI[c1], mI[c2], n);
            } else if (g_rfo_variants & RFO_VAR_GF_DIAG_ADD_EPS) {
                ew_sub_mul(cv, mI[c1], mI[c2], n);
                for (size_t i = 0; i < n; i++)
                    cv[i] = cv[i] + diag;
            } else {
                ew_sub_mad(cv, mI[c1], mI[c2], -diag, n);
            }
        }
    /* ComputeCovGuideInv_ParBody, 3-channel branch */
    float *det = (float *)malloc(sizeof(float) * n);
    for (int k = 0; k < 3; k++)
        for (int l = 0; l <= k; l++) {
            float *dv = inv[sym_idx(k, l)];
            const float *a00 = cov[sym_idx((k + 1) % 3, (l + 1) % 3)];
            const float *a01 = cov[sym_idx((k + 1) % 3, (l + 2) % 3)];
            const float *a10 = cov[sym_idx((k + 2) % 3, (l + 1) % 3)];
            const float *a11 = cov[sym_idx((k + 2) % 3, (l + 2) % 3)];
            ew_mul(dv, a00, a11, n);
            ew_sub_mul(dv, a01, a10, n);
        }
    for (int k = 0; k < 3; k++) {
        const float *a = cov[sym_idx(k, 0)];
        const float *ac = inv[sym_idx(k, 0)];
        if (k == 0)
            ew_mul(det, a, ac, n);
        else
            ew_add_mul(det, a, ac, n);
    }
    if (eps < 1e-2)
        for (size_t i = 0; i < n; i++)
            if (fabsf(det[i]) < 1e-6f)
                det[i] = 1.f;
    for (int k = 0; k < 6; k++)
        for (size_t i = 0; i < n; i++)
            inv[k][i] = inv[k][i] / det[i];
    free(det);
    for (int k = 0; k < 6; k++)
        free(cov[k]);

    /* ---- filter(src) ---- */
    float *p[3] = {0, 0, 0}, *cp[3][3], *al[3][3];
    for (int s = 0; s < src_cn; s++) {
        p[s] = (float *)malloc(sizeof(float) * n);
        for (size_t i = 0; i < n; i++)
            p[s][i] = src_f[i * src_cn + s];
        for (int g = 0; g < 3; g++) {
            cp[s][g] = (float *)malloc(sizeof(float) * n);
            al[s][g] = (float *)malloc(sizeof(float) * n);
            ew_mul(cp[s][g], p[s], I[g], n);
        }
    }
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (int t = 0; t < src_cn * 4; t++) {
        int s = t / 4, g = t % 4;
        if (g == 3)
            rfo_box_mean(p[s], p[s], h, w, radius); /* srcCnMean aliases srcCn */
        else
            rfo_box_mean(cp[s][g], cp[s][g], h, w, radius);
    }
    for (int s = 0; s < src_cn; s++)
        for (int g = 0; g < 3; g++)
            ew_sub_mul(cp[s][g], p[s], mI[g], n);
    for (int s = 0; s < src_cn; s++)
        for (int g = 0; g < 3; g++)
            for (int k = 0; k < 3; k++) {
                const float *A = inv[sym_idx(g, k)];
                if (k == 0)
                    ew_mul(al[s][g], A, cp[s][k], n);
                else
                    ew_add_mul(al[s][g], A, cp[s][k], n);
            }
    /* beta aliases srcCnMean: beta -= alpha_g * mean_g, g ascending */
    for (int s = 0; s < src_cn; s++)
        for (int g = 0; g < 3; g++)
            ew_sub_mul(p[s], al[s][g], mI[g], n);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (int t = 0; t < src_cn * 4; t++) {
        int s = t / 4, g = t % 4;
        float *pl = g == 3 ? p[s] : al[s][g];
        if (g_rfo_census_on && !(g_rfo_variants & RFO_VAR_GF_FLOAT_BOXSUM))
            rfo_box_mean_census(pl, pl, h, w, radius); /* same values, operations counted */
        else
            rfo_box_mean(pl, pl, h, w, radius);
    }
    /* ApplyTransform: q = beta + sum_g alpha_g * I_g, g ascending */
    for (int s = 0; s < src_cn; s++)
        for (int g = 0; g < 3; g++)
            ew_add_mul(p[s], al[s][g], I[g], n);
    for (int s = 0; s < src_cn; s++)
        for (size_t i = 0; i < n; i++) {
            if (q_f32)
                q_f32[i * src_cn + s] = p[s][i];
            if (dst)
                dst[i * src_cn + s] = rfo_saturate_u8(p[s][i]);
        }
    for (int s = 0; s < src_cn; s++) {
        free(p[s]);
        for (int g = 0; g < 3; g++) {
            free(cp[s][g]);
            free(al[s][g]);
        }
    }
    for (int c = 0; c < 3; c++) {
        free(I[c]);
        free(mI[c]);
    }
    for (int k = 0; k < 6; k++)
        free(inv[k]);
    return 0;
}

int rfo_gf_u8(const uint8_t *guide, const uint8_t *src, uint8_t *dst, float *q_f32, int h, int w,
              int guide_cn, int src_cn, int radius, double eps, int threads)
{
    if (!guide || !src || (!dst && !q_f32) || h <= 0 || w <= 0 || radius < 0)
        return -1;
    if (guide_cn != 3 || (src_cn != 1 && src_cn != 3))
        return -1;
    size_t n = (size_t)h * w;
    float *gf = (float *)malloc(sizeof(float) * n * 3);
    float *sf = (float *)malloc(sizeof(float) * n * src_cn);
    for (size_t i = 0; i < n * 3; i++)
        gf[i] = (float)guide[i];
    for (size_t i = 0; i < n * src_cn; i++)
        sf[i] = (float)src[i];
    int rc = gf_core(gf, sf, dst, q_f32, h, w, src_cn, radius, eps, threads);
    free(gf);
    free(sf);
    return rc;
}

/* guidedFilter on CV_32F guide and src (dDepth = -1: float result, no rounding). */
int rfo_gf_f32(const float *guide, const float *src, float *dst, int h, int w, int guide_cn,
               int src_cn, int radius, double eps, int threads)
{
    if (!guide || !src || !dst || h <= 0 || w <= 0 || radius < 0)
        return -1;
    if (guide_cn != 3 || (src_cn != 1 && src_cn != 3))
        return -1;
    return gf_core(guide, src, NULL, dst, h, w, src_cn, radius, eps, threads);
}

/* The census form of the box mean alone (tests/test_oracle.py); counters through rfo_census. */
void rfo_box_census_f32(const float *src, float *dst, int h, int w, int r)
{
    rfo_box_mean_census(src, dst, h, w, r);
}

/* Exposed for unit tests of the box mean alone. */
void rfo_box_mean_f32(const float *src, float *dst, int h, int w, int r)
{
    rfo_box_mean(src, dst, h, w, r);
}

/* ------------------------------------------------------------------------- */
/* 1x1 CNN (network_definition.prototxt:9-165, learned_weights.caffemodel)   */
/* ------------------------------------------------------------------------- */

#define RFO_CNN_NPARAMS 4513
/* weights layout (4513 floats): W0[32][3] b0[32] | W1[32][32] b1[32] | ... W4 b4 |
 * wf[160] bf[1]  -- the order the blobs appear in the caffemodel. */

/* sRGB byte -> linear float32 exactly as imgCV2_to_caffeBlob + image_utils.srgb_to_rgb
 * do in float64, then the float32 blob assignment (decompose_with_trained_CNN.py:60-68,88). */
void rfo_srgb_lut(float *lut256)
{
    for (int v = 0; v < 256; v++) {
        double s = (double)v / 255.0;
        double lin = s <= 0.04045 ? s / 12.92 : pow((s + 0.055) / 1.055, 2.4);
        lut256[v] = (float)lin;
    }
}

/*
 * Caffe forward of the shipped net on one uint8 BGR image (h*w*3).
 * conv = sgemm over k ascending (fused multiply-add chain from 0, the order of
 * a BLAS micro-kernel on FMA hardware) followed by the bias gemm (one add);
 * ReLU in place; concat order conv0..conv4; sigmoid = 1/(1+exp(-x)) evaluated
 * as caffe's `1. / (1. + exp(-x))` (float exp, double division).
 * r: h*w float32 (blob reflectance_intensity); r_u8 (optional): the `-r.png`
 * bytes = (r*255).astype(uint8), i.e. float32 multiply then truncation
 * (image_utils.py:63-68; normalize() is the identity for sigmoid outputs).
 */
int rfo_cnn_reflectance_u8(const uint8_t *bgr, float *r, uint8_t *r_u8, int h, int w,
                           const float *weights, int threads)
{
    if (!bgr || !weights || (!r && !r_u8) || h <= 0 || w <= 0)
        return -1;
    float lut[256];
    rfo_srgb_lut(lut);
    const float *W0 = weights, *b0 = weights + 96;
    const float *Wl[4], *bl[4];
    const float *q = weights + 128;
    for (int l = 0; l < 4; l++) {
        Wl[l] = q;
        bl[l] = q + 1024;
        q += 1056;
    }
    const float *wf = q, *bf = q + 160;
    size_t n = (size_t)h * w;
    const int no_fma = (g_rfo_variants & RFO_VAR_CNN_GEMM_NO_FMA) != 0;
    const int sig_tanh = (g_rfo_variants & RFO_VAR_CNN_SIGMOID_TANH) != 0;
#define RFO_MAC(wv, xv, acc) (no_fma ? (acc) + (wv) * (xv) : fmaf((wv), (xv), (acc)))
#ifdef _OPENMP
    if (threads <= 0)
        threads = omp_get_max_threads();
#pragma omp parallel for num_threads(threads)
#endif
    for (size_t i = 0; i < n; i++) {
        /* blob channel order is RGB: blob[:, :, ::-1] of the BGR image */
        float x[3] = {lut[bgr[i * 3 + 2]], lut[bgr[i * 3 + 1]], lut[bgr[i * 3 + 0]]};
        float hcur[32], hnext[32], cat[160];
        for (int o = 0; o < 32; o++) {
            float acc = 0.0f;
            for (int k = 0; k < 3; k++)
                acc = RFO_MAC(W0[o * 3 + k], x[k], acc);
            acc = acc + b0[o];
            hcur[o] = acc > 0.0f ? acc : 0.0f;
            cat[o] = hcur[o];
        }
        for (int l = 0; l < 4; l++) {
            for (int o = 0; o < 32; o++) {
                float acc = 0.0f;
                for (int k = 0; k < 32; k++)
                    acc = RFO_MAC(Wl[l][o * 32 + k], hcur[k], acc);
                acc = acc + bl[l][o];
                hnext[o] = acc > 0.0f ? acc : 0.0f;
            }
            memcpy(hcur, hnext, sizeof(hcur));
            memcpy(cat + 32 * (l + 1), hcur, sizeof(hcur));
        }
        float z = 0.0f;
        for (int k = 0; k < 160; k++)
            z = RFO_MAC(wf[k], cat[k], z);
        z = z + bf[0];
        /* expf(-z) modelled as the correctly rounded float of exp in double */
        float e = (float)exp((double)(-z));
        float rv = (float)(1.0 / (1.0 + (double)e));
        if (sig_tanh) /* 0.5 * tanh(0.5 * x) + 0.5 in float, tanhf modelled as rounded double tanh */
            rv = 0.5f * (float)tanh((double)(0.5f * z)) + 0.5f;
        if (r)
            r[i] = rv;
        if (r_u8) {
            float s = rv * 255.0f;
            r_u8[i] = (uint8_t)s; /* astype(uint8): truncation toward zero */
        }
    }
#undef RFO_MAC
    return 0;
}
