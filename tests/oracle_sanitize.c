/* Sanitizer harness for the CPU oracle (test infrastructure): builds oracle/rf_oracle.c with
 * -fsanitize=address,undefined and drives every entry point on small odd-shaped inputs, incl.
 * images narrower than the radius (multi-bounce borders).  Run by tests/test_oracle_sanitize.py.
 *   gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fopenmp \
 *       tests/oracle_sanitize.c oracle/rf_oracle.c -lm -o /tmp/oracle_sanitize && /tmp/oracle_sanitize
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int rfo_jbf_u8(const uint8_t *joint, const uint8_t *src, uint8_t *dst, int h, int w, int joint_cn,
               int src_cn, int d, double sigma_color, double sigma_space, int border, int flags,
               int threads);
int rfo_jbf_f32(const float *joint, const float *src, float *dst, int h, int w, int joint_cn,
                int src_cn, int d, double sigma_color, double sigma_space, int border, int threads);
int rfo_gf_u8(const uint8_t *guide, const uint8_t *src, uint8_t *dst, float *q_f32, int h, int w,
              int guide_cn, int src_cn, int radius, double eps, int threads);
int rfo_gf_f32(const float *guide, const float *src, float *dst, int h, int w, int guide_cn,
               int src_cn, int radius, double eps, int threads);
int rfo_cnn_reflectance_u8(const uint8_t *bgr, float *r, uint8_t *r_u8, int h, int w,
                           const float *weights, int threads);
void rfo_box_mean_f32(const float *src, float *dst, int h, int w, int r);

static uint32_t rng = 12345u;
static uint32_t next(void) { rng = rng * 1664525u + 1013904223u; return rng >> 8; }

int main(void)
{
    static const int shapes[][2] = {{1, 1}, {3, 5}, {7, 40}, {33, 17}, {64, 65}};
    int fails = 0;
    for (unsigned si = 0; si < sizeof(shapes) / sizeof(shapes[0]); si++) {
        const int h = shapes[si][0], w = shapes[si][1];
        const size_t n = (size_t)h * w;
        uint8_t *g = malloc(n * 3), *s3 = malloc(n * 3), *s1 = malloc(n), *d = malloc(n * 3);
        float *gf = malloc(n * 3 * 4), *sf = malloc(n * 3 * 4), *df = malloc(n * 3 * 4);
        for (size_t i = 0; i < n * 3; i++) {
            g[i] = (uint8_t)next();
            s3[i] = (uint8_t)next();
            gf[i] = g[i] / 255.f;
            sf[i] = s3[i] / 255.f;
        }
        for (size_t i = 0; i < n; i++)
            s1[i] = (uint8_t)next();
        for (int border = 0; border <= 4; border++) {
            fails += rfo_jbf_u8(g, s3, d, h, w, 3, 3, -1, 20.0, 22.0, border, 0, 2) != 0;
            fails += rfo_jbf_u8(s1, s1, d, h, w, 1, 1, 9, 20.0, 3.0, border, 1, 1) != 0;
        }
        fails += rfo_jbf_u8(g, s1, d, h, w, 3, 1, -1, 15.0, 28.0, 4, 0, 2) != 0;
        if (n > 1) {  /* a constant joint is refused by the float variant */
            fails += rfo_jbf_f32(gf, sf, df, h, w, 3, 3, -1, 0.08, 5.0, 4, 2) != 0;
            fails += rfo_jbf_f32(gf, sf, df, h, w, 3, 3, 7, 0.08, 2.0, 1, 1) != 0;
        }
        fails += rfo_gf_u8(g, s3, d, df, h, w, 3, 3, 45, 3.0, 2) != 0;
        fails += rfo_gf_u8(g, s1, d, NULL, h, w, 3, 1, 52, 7.0, 1) != 0;
        fails += rfo_gf_u8(g, s3, d, NULL, h, w, 3, 3, 0, 1e-3, 1) != 0;
        fails += rfo_gf_f32(gf, sf, df, h, w, 3, 3, 9, 1e-4, 2) != 0;
        rfo_box_mean_f32(sf, df, h, w, 4);
        free(g); free(s3); free(s1); free(d); free(gf); free(sf); free(df);
    }
    {
        /* CNN: 3->32, 4 x 32->32, 160->1 + biases = 4513 floats */
        const int h = 5, w = 9;
        float *wts = malloc(4513 * 4), *r = malloc((size_t)h * w * 4);
        uint8_t *img = malloc((size_t)h * w * 3), *r8 = malloc((size_t)h * w);
        for (int i = 0; i < 4513; i++)
            wts[i] = ((int)(next() & 0xffff) - 32768) / 200000.f;
        for (int i = 0; i < h * w * 3; i++)
            img[i] = (uint8_t)next();
        fails += rfo_cnn_reflectance_u8(img, r, r8, h, w, wts, 2) != 0;
        free(wts); free(r); free(img); free(r8);
    }
    printf("oracle_sanitize: %d failing call(s)\n", fails);
    return fails != 0;
}
