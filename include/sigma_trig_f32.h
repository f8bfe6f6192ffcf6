/* sigma_trig_f32.h -- the fp32 sin / cos / tan / atan / atan2 of the arithmetic contract, shared by the HIP kernels
 * (sigmarl_amd/csrc) and the CPU oracle (oracle/).
 *
 * What the reference's arithmetic is (measured in the build container, tools/torch_trig_probe.py): PyTorch-CPU evaluates float32
 *   - atan2 with SLEEF's 1.0-ULP function (Sleef_atan2f16_u10 of the SLEEF 3.6 bundled with torch 2.10; 0 differing results of 4e6),
 *   - sin / cos / tan / atan with something else: neither the u10 nor the u35 SLEEF functions exported by libtorch_cpu.so reproduce
 *     torch.sin / cos / tan / atan (1.9 % / 2.2 % / 12 % / 1.8 % of the results differ from the u10 functions), the results are within
 *     one ulp of the correctly rounded value (sin / cos differ from it in 4.9 % of the cases, tan in 0.6 %, atan in 0.05 %): the MKL
 *     vector math library the build links (closed source, not restatable).
 * Contract:
 *   - sin / cos / tan / atan: the CORRECTLY ROUNDED fp32 value, obtained as (float) of an fp64 evaluation.  Both sides run the same
 *     fp64 algorithm below (3-term Cody-Waite reduction by pi/2, the sine / cosine / arctangent kernels of fdlibm in Horner form; every
 *     operation an IEEE fp64 operation or an explicit fma), so host and device agree bit for bit by construction; its error is about one
 *     fp64 ulp, i.e. the result is the correctly rounded fp32 value for all but ~1e-8 of the arguments (0 of 4e6 sampled arguments per
 *     function differ from (float)libm(double), tests/test_trig.py).  Within one ulp of torch.  Arguments beyond 2^30 return NaN.
 *   - atan2: SLEEF's published algorithm (xatan2f_u1 / atan2kf_u1 of sleefsimdsp.c with the FMA double-float helpers of df.h) restated
 *     operation by operation: the SAME BITS as torch.atan2 (pinned by tests/golden/trig_f32.npz, generator
 *     tests/golden/gen/gen_trig_golden.py).
 *
 * Compiles as C99 (gcc -mfma -ffp-contract=off) and as HIP device code (-ffp-contract=off): every fused operation is an explicit
 * fma / fmaf, every other operation a single IEEE operation; `/` is the correctly rounded division on both sides.
 */
#ifndef SIGMA_TRIG_F32_H
#define SIGMA_TRIG_F32_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__HIP_DEVICE_COMPILE__)
#define SIGMA_TRIG_FN __host__ __device__ static __forceinline__
#else
#define SIGMA_TRIG_FN static inline
#endif

typedef struct { float x, y; } sigma_df_t;

SIGMA_TRIG_FN uint32_t sigma_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
SIGMA_TRIG_FN float sigma_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
SIGMA_TRIG_FN float sigma_mulsign(float x, float y) { return sigma_u2f(sigma_f2u(x) ^ (sigma_f2u(y) & 0x80000000u)); }

/* ---- double-float helpers (sleef/src/libm/df.h, the ENABLE_FMA_SP forms) ------------------------------------------ */
SIGMA_TRIG_FN sigma_df_t sigma_df(float x, float y) { sigma_df_t r; r.x = x; r.y = y; return r; }
SIGMA_TRIG_FN sigma_df_t sigma_dfnormalize(sigma_df_t t) { float s = t.x + t.y; return sigma_df(s, (t.x - s) + t.y); }
SIGMA_TRIG_FN sigma_df_t sigma_dfneg(sigma_df_t t) { return sigma_df(-t.x, -t.y); }
/* |x| >= |y| */
SIGMA_TRIG_FN sigma_df_t sigma_dfadd_f_f(float x, float y) { float s = x + y; return sigma_df(s, (x - s) + y); }
SIGMA_TRIG_FN sigma_df_t sigma_dfadd2_f_f(float x, float y) {
  float s = x + y, v = s - x;
  return sigma_df(s, (x - (s - v)) + (y - v));
}
SIGMA_TRIG_FN sigma_df_t sigma_dfadd_f2_f(sigma_df_t x, float y) { float s = x.x + y; return sigma_df(s, ((x.x - s) + y) + x.y); }
SIGMA_TRIG_FN sigma_df_t sigma_dfadd2_f2_f(sigma_df_t x, float y) {
  float s = x.x + y, v = s - x.x;
  float t = (x.x - (s - v)) + (y - v);
  return sigma_df(s, t + x.y);
}
SIGMA_TRIG_FN sigma_df_t sigma_dfadd_f_f2(float x, sigma_df_t y) { float s = x + y.x; return sigma_df(s, ((x - s) + y.x) + y.y); }
SIGMA_TRIG_FN sigma_df_t sigma_dfadd_f2_f2(sigma_df_t x, sigma_df_t y) {
  float s = x.x + y.x;
  return sigma_df(s, (((x.x - s) + y.x) + x.y) + y.y);
}
SIGMA_TRIG_FN sigma_df_t sigma_dfmul_f2_f2(sigma_df_t x, sigma_df_t y) {
  float s = x.x * y.x;
  return sigma_df(s, fmaf(x.x, y.y, fmaf(x.y, y.x, fmaf(x.x, y.x, -s))));
}
SIGMA_TRIG_FN float sigma_dfmul_f_f2_f2(sigma_df_t x, sigma_df_t y) { return fmaf(x.x, y.x, fmaf(x.y, y.x, x.x * y.y)); }
SIGMA_TRIG_FN sigma_df_t sigma_dfmul_f2_f(sigma_df_t x, float y) {
  float s = x.x * y;
  return sigma_df(s, fmaf(x.y, y, fmaf(x.x, y, -s)));
}
SIGMA_TRIG_FN sigma_df_t sigma_dfsqu(sigma_df_t x) {
  float s = x.x * x.x;
  return sigma_df(s, fmaf(x.x + x.x, x.y, fmaf(x.x, x.x, -s)));
}
SIGMA_TRIG_FN sigma_df_t sigma_dfrec_f2(sigma_df_t d) {
  float s = 1.0f / d.x;
  return sigma_df(s, s * fmaf(-d.y, s, fmaf(-d.x, s, 1.0f)));
}
SIGMA_TRIG_FN sigma_df_t sigma_dfdiv(sigma_df_t n, sigma_df_t d) {
  float t = 1.0f / d.x;
  float s = n.x * t;
  float u = fmaf(t, n.x, -s);
  float v = fmaf(-d.y, t, fmaf(-d.x, t, 1.0f));
  return sigma_df(s, fmaf(s, v, fmaf(n.y, t, u)));
}



/* ---- sin / cos in fp64: 3-term Cody-Waite reduction by pi/2 (exact products for |n| < 2^20, graceful beyond), the kernels of fdlibm
 * k_sin.c / k_cos.c in Horner form.  Arguments beyond 2^30 (and NaN / inf) return NaN. ---- */
SIGMA_TRIG_FN void sigma_sincos_f64(double x, double* sn, double* cs) {
  if (!(fabs(x) < 1073741824.0)) { *sn = NAN; *cs = NAN; return; }
  if (x == 0.0) { *sn = x; *cs = 1.0; return; }  /* keeps the sign of a zero argument */
  const double n = rint(x * 6.36619772367581382433e-01);
  double r = fma(-n, 1.57079632673412561417e+00, x);
  r = fma(-n, 6.07710050630396597660e-11, r);
  r = fma(-n, 2.02226624879595063154e-21, r);
  const double z = r * r;
  double ps = 1.58969099521155010221e-10;
  ps = fma(ps, z, -2.50507602534068634195e-08);
  ps = fma(ps, z, 2.75573137070700676789e-06);
  ps = fma(ps, z, -1.98412698298579493134e-04);
  ps = fma(ps, z, 8.33333333332248946124e-03);
  ps = fma(ps, z, -1.66666666666666324348e-01);
  const double s = fma(r * z, ps, r);
  double pc = -1.13596475577881948265e-11;
  pc = fma(pc, z, 2.08757232129817482790e-09);
  pc = fma(pc, z, -2.75573143513906633035e-07);
  pc = fma(pc, z, 2.48015872894767294178e-05);
  pc = fma(pc, z, -1.38888888888741095749e-03);
  pc = fma(pc, z, 4.16666666666666019037e-02);
  const double c = fma(z * z, pc, fma(z, -0.5, 1.0));
  const double k = n - 4.0 * floor(n * 0.25);  /* n mod 4 in {0, 1, 2, 3} */
  const double s1 = (k == 1.0 || k == 3.0) ? c : s, c1 = (k == 1.0 || k == 3.0) ? s : c;
  *sn = (k == 2.0 || k == 3.0) ? -s1 : s1;
  *cs = (k == 1.0 || k == 2.0) ? -c1 : c1;
}

/* atan in fp64: the argument reduction and the odd polynomial of fdlibm s_atan.c (atan(x) = atanhi[id] + atan((x - c) / (1 + c x)),
 * c in {0.5, 1, 1.5, inf}; |reduced x| < 7/16), Horner steps as explicit fma. */
SIGMA_TRIG_FN double sigma_atan_f64(double xx) {
  const double ax = fabs(xx);
  if (ax != ax) return xx;
  double x = ax, hi = 0.0, lo = 0.0;
  int red = 1;
  if (ax < 0.4375) { red = 0; }
  else if (ax < 0.6875) { x = (2.0 * ax - 1.0) / (2.0 + ax); hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17; }
  else if (ax < 1.1875) { x = (ax - 1.0) / (ax + 1.0); hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17; }
  else if (ax < 2.4375) { x = (ax - 1.5) / (1.0 + 1.5 * ax); hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17; }
  else { x = -1.0 / ax; hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17; }
  const double z = x * x, w = z * z;
  double s1 = 1.62858201153657823623e-02;
  s1 = fma(s1, w, 4.97687799461593236017e-02);
  s1 = fma(s1, w, 6.66107313738753120669e-02);
  s1 = fma(s1, w, 9.09088713343650656196e-02);
  s1 = fma(s1, w, 1.42857142725034663711e-01);
  s1 = fma(s1, w, 3.33333333333329318027e-01);
  s1 = s1 * z;
  double s2 = -3.65315727442169155270e-02;
  s2 = fma(s2, w, -5.83357013379057348645e-02);
  s2 = fma(s2, w, -7.69187620504482999495e-02);
  s2 = fma(s2, w, -1.11111104054623557880e-01);
  s2 = fma(s2, w, -1.99999999998764832476e-01);
  s2 = s2 * w;
  double r;
  if (!red) r = x - x * (s1 + s2);
  else r = hi - ((x * (s1 + s2) - lo) - x);
  return copysign(r, xx);
}

/* ---- the contract's sin / cos / tan / atan: (float) of the fp64 value ---------------------------------------------------- */
SIGMA_TRIG_FN void sigma_sincosf(float d, float* sn, float* cs) { double s, c; sigma_sincos_f64((double)d, &s, &c); *sn = (float)s; *cs = (float)c; }
SIGMA_TRIG_FN float sigma_sinf(float d) { double s, c; sigma_sincos_f64((double)d, &s, &c); return (float)s; }
SIGMA_TRIG_FN float sigma_cosf(float d) { double s, c; sigma_sincos_f64((double)d, &s, &c); return (float)c; }
SIGMA_TRIG_FN float sigma_tanf(float d) { double s, c; sigma_sincos_f64((double)d, &s, &c); return (float)(s / c); }
SIGMA_TRIG_FN float sigma_atanf(float d) { return (float)sigma_atan_f64((double)d); }

/* atan2kf_u1: atan(y / x) as a double-float, for y >= 0 */
SIGMA_TRIG_FN sigma_df_t sigma_atan2kf_u1(sigma_df_t y, sigma_df_t x) {
  int q = 0;
  if (x.x < 0.0f) { q = -2; x = sigma_dfneg(x); }
  sigma_df_t s, t;
  if (x.x < y.x) { q += 1; s = sigma_dfneg(x); t = y; }
  else { s = y; t = x; }
  s = sigma_dfdiv(s, t);
  t = sigma_dfsqu(s);
  t = sigma_dfnormalize(t);
  float u = -0.00176397908944636583328247f;
  u = fmaf(u, t.x, 0.0107900900766253471374512f);
  u = fmaf(u, t.x, -0.0309564601629972457885742f);
  u = fmaf(u, t.x, 0.0577365085482597351074219f);
  u = fmaf(u, t.x, -0.0838950723409652709960938f);
  u = fmaf(u, t.x, 0.109463557600975036621094f);
  u = fmaf(u, t.x, -0.142626821994781494140625f);
  u = fmaf(u, t.x, 0.199983194470405578613281f);
  t = sigma_dfmul_f2_f2(t, sigma_dfadd_f_f(-0.333332866430282592773438f, u * t.x));
  t = sigma_dfmul_f2_f2(s, sigma_dfadd_f_f2(1.0f, t));
  t = sigma_dfadd_f2_f2(sigma_dfmul_f2_f(sigma_df(1.5707963705062866211f, -4.3711388286737928865e-08f), (float)q), t);
  return t;
}

/* Sleef_atan2f*_u10 (xatan2f_u1) */
SIGMA_TRIG_FN float sigma_atan2f(float y, float x) {
  const float FLT_MIN_ = 1.17549435082228750797e-38f;
  if (fabsf(x) < 2.9387372783541830947e-39f) { y *= (float)(1ULL << 24); x *= (float)(1ULL << 24); }  /* underflow guard of xatan2f_u1 */
  sigma_df_t d = sigma_atan2kf_u1(sigma_df(fabsf(y), 0.0f), sigma_df(x, 0.0f));
  float r = d.x + d.y;
  (void)FLT_MIN_;
  r = sigma_mulsign(r, x);
  if (isinf(x) || x == 0.0f) r = 1.570796326794896557998982f - (isinf(x) ? (sigma_mulsign(1.0f, x) * 1.570796326794896557998982f) : 0.0f);
  if (isinf(y)) r = 1.570796326794896557998982f - (isinf(x) ? (sigma_mulsign(1.0f, x) * (float)(3.14159265358979323846 / 4)) : 0.0f);
  if (y == 0.0f) r = (sigma_f2u(x) >> 31) ? 3.14159265358979323846f : 0.0f;
  if (isnan(x) || isnan(y)) return NAN;
  return sigma_mulsign(r, y);
}

#endif /* SIGMA_TRIG_F32_H */
