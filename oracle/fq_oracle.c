/*
 * fq_oracle.c - scalar C restatement of the reference's fake-quantize arithmetic.
 * TEST INFRASTRUCTURE ONLY (the checker, never the product; see oracle/fakequant_oracle.py).
 *
 * Independent of ATen: plain IEEE fp32 C (build with -O2 -ffp-contract=off, no -ffast-math), one
 * function per reference expression.  Parity status: PINNED - tests/test_oracle_c.py checks every
 * function against tests/golden/golden_v1.npz (vectors produced by the reference's own code).
 * Citations are relative to /root/reference.
 */
#include <math.h>
#include <stdint.h>

static float clamp_t(float v, float lo, float hi) { /* torch.clamp: NaN propagates */
  return v < lo ? lo : (v > hi ? hi : v);
}
static float relu_t(float v) { return v < 0.0f ? 0.0f : v; }
static float ste_round(float v) { /* utils.py:29-32 forward value */
  float r = rintf(v);
  return (r - v) + v;
}
static float ste_floor(float v) { /* utils.py:34-37 forward value */
  float r = floorf(v);
  return (r - v) + v;
}
static float ste_scale(float s, float g) { /* utils.py:24-27 forward value */
  float sg = s * g;
  return (s - sg) + sg;
}
static int64_t chan(int64_t i, int64_t channels, int64_t inner) {
  return channels == 1 ? 0 : (i / inner) % channels;
}

/* form 0: utils.py:1-11 emulate_quantize */
void fqo_emulate(const float* x, float* q, float* y, const float* s, const float* o, int64_t n, int64_t channels,
                 int64_t inner, float lo, float hi) {
  for (int64_t i = 0; i < n; ++i) {
    int64_t c = chan(i, channels, inner);
    float qq = clamp_t(rintf((x[i] - o[c]) / (s[c] + 1e-7f)), lo, hi);
    q[i] = qq;
    y[i] = qq * s[c] + o[c];
  }
}

/* form 1: modules/base.py:96-102,131-133 */
void fqo_qbase(const float* x, float* q, float* y, const float* s, const float* o, int64_t n, int64_t channels,
               int64_t inner, float lo, float hi, float g) {
  for (int64_t i = 0; i < n; ++i) {
    int64_t c = chan(i, channels, inner);
    float sh = ste_scale(s[c], g);
    float qq = ste_round(clamp_t((x[i] - o[c]) / sh, lo, hi));
    q[i] = qq;
    y[i] = qq * sh + o[c];
  }
}

/* form 2: FSPTQuant/base.py:108-109 */
void fqo_zeropoint(const float* x, float* q, float* y, const float* s, const float* zp, int64_t n, int64_t channels,
                   int64_t inner, float lo, float hi) {
  for (int64_t i = 0; i < n; ++i) {
    int64_t c = chan(i, channels, inner);
    float qq = clamp_t(ste_round(x[i] / s[c]) + zp[c], lo, hi);
    q[i] = qq;
    y[i] = (qq - zp[c]) * s[c];
  }
}

/* form 3: FSPTQuant/base.py:149-152 */
void fqo_symmetric(const float* x, float* q, float* y, const float* s, int64_t n, int64_t channels, int64_t inner,
                   float lo, float hi) {
  for (int64_t i = 0; i < n; ++i) {
    int64_t c = chan(i, channels, inner);
    float qq = clamp_t(ste_round(x[i] / s[c]), lo, hi);
    q[i] = qq;
    y[i] = qq * s[c];
  }
}

/* form 4: RootQ/base.py:106-111 + RootQ/function.py:15-20 */
void fqo_rootq_act(const float* x, float* q, float* y, const float* s, int64_t n, int64_t channels, int64_t inner,
                   float lo, float hi) {
  for (int64_t i = 0; i < n; ++i) {
    int64_t c = chan(i, channels, inner);
    float up = s[c] * (hi - lo);
    float t = x[i] + relu_t(0.0f - x[i]);
    t = t - relu_t(t - up);
    float qq = ste_round(t / s[c]);
    q[i] = qq;
    y[i] = qq * s[c];
  }
}

/* RootQ/base.py:146-155 + RootQ/function.py:22-32,58-67; sgn(phi) evaluated as the reference does */
void fqo_rootq_weight(const float* w, float* y, int64_t n, float up, float lw, float alpha, float lo, float hi) {
  float delta = (up - lw) / (hi - lo);
  float a = alpha + relu_t(1e-4f - alpha);
  a = a - relu_t(a - 1.0f);
  for (int64_t i = 0; i < n; ++i) {
    float t = w[i] + relu_t(lw - w[i]);
    t = t - relu_t(t - up);
    float iv = ste_floor((t - lw) / delta);
    float mi = (iv + 0.5f) * delta + lw;
    float d = t - mi;
    float sg = d / (fabsf(d) + 1e-5f);
    float phi = powf((2.0f / delta) * fabsf(d) + 1e-5f, a) * sg;
    float s = (float)((0.0f < phi) - (phi < 0.0f));
    y[i] = ((s + 1.0f) / 2.0f + iv) * delta + lw;
  }
}

/* ops.py:20-34 / :112-140: per-channel (channels = 1: per tensor) max, min, max|x|; NaN propagates */
void fqo_minmax(const float* x, float* vmax, float* vmin, float* vabs, int64_t outer, int64_t channels, int64_t inner) {
  for (int64_t c = 0; c < channels; ++c) {
    float mx = -INFINITY, mn = INFINITY, ab = 0.0f;
    int nan = 0;
    for (int64_t o = 0; o < outer; ++o)
      for (int64_t i = 0; i < inner; ++i) {
        float v = x[(o * channels + c) * inner + i];
        if (v != v) nan = 1;
        if (v > mx) mx = v;
        if (v < mn) mn = v;
        if (fabsf(v) > ab) ab = fabsf(v);
      }
    vmax[c] = nan ? NAN : mx;
    vmin[c] = nan ? NAN : mn;
    vabs[c] = nan ? NAN : ab;
  }
}

/* the scale/offset arithmetic of ops.py:22-24,26-33 */
void fqo_qparams(const float* vmax, const float* vmin, const float* vabs, float* scale, float* offset, int64_t channels,
                 int n_bits, int is_signed, int allow_offset, float eps) {
  for (int64_t c = 0; c < channels; ++c) {
    if (is_signed) {
      scale[c] = vabs[c] / (float)((1 << (n_bits - 1)) - 1);
      offset[c] = 0.0f;
    } else {
      float mn = allow_offset ? vmin[c] : 0.0f;
      scale[c] = (vmax[c] - mn) / (float)((1 << n_bits) - 1);
      offset[c] = mn;
    }
    if (eps != 0.0f) scale[c] = scale[c] + eps;
  }
}

/* autograd through modules/base.py:96-102, node by node (see qbase_backward in fakequant_oracle.py) */
void fqo_qbase_backward(const float* x, const float* gy, float* gx, double* gscale_sum, float s, float o, int64_t n,
                        float lo, float hi, float g) {
  float sh = ste_scale(s, g);
  double acc = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    float v = (x[i] - o) / sh;
    float r = ste_round(clamp_t(v, lo, hi));
    int inside = (v >= lo) && (v <= hi);
    float gv = inside ? gy[i] * sh : 0.0f;
    gx[i] = gv / sh;
    acc += (double)(gy[i] * r) + (double)((-gv) * (v / sh));
  }
  *gscale_sum = acc;
}

/* two 4-bit codes per byte, element 2i in the low nibble (the build's own layout: DLMCQ_CODES_P4) */
void fqo_pack4(const int8_t* codes, uint8_t* packed, int64_t n) {
  for (int64_t b = 0; b < (n + 1) / 2; ++b) {
    uint8_t l = (uint8_t)codes[2 * b] & 0xf;
    uint8_t h = (2 * b + 1 < n) ? ((uint8_t)codes[2 * b + 1] & 0xf) : 0;
    packed[b] = (uint8_t)(l | (h << 4));
  }
}
