/* sigma_trig_f32.h -- the fp32 sin / cos / tan / atan of the arithmetic contract, shared by the HIP kernels
 * (sigmarl_amd/csrc) and the CPU oracle (oracle/).
 *
 * What the reference's arithmetic is (measured in the build container, tools/torch_trig_probe.py): PyTorch-CPU evaluates float32
 * sin / cos / tan / atan with a vector math library that is neither of the SLEEF functions libtorch_cpu.so exports (1.9 % / 2.2 % / 12 % /
 * 1.8 % of torch's results differ from Sleef_{sin,cos,tan,atan}f16_u10, more from the u35 forms) -- the MKL vector math the build links,
 * closed source and not restatable.  Its results are within one ulp of the correctly rounded value (sin / cos differ from it in 4.9 % of
 * the cases, tan in 0.6 %, atan in 0.05 %).  (atan2, which only the oracle's observation needs, is SLEEF's u10 function in the vectorised
 * body of a tensor and glibc's atan2f in its tail of < 16 elements -- the oracle uses the correctly rounded value there as well.)
 *
 * Contract: sin / cos / tan / atan are the CORRECTLY ROUNDED fp32 value, obtained as (float) of an fp64 evaluation.  Both sides run the
 * same fp64 algorithm below (3-term Cody-Waite reduction by pi/2, the sine / cosine / arctangent kernels of fdlibm in Horner form; every
 * operation an IEEE fp64 operation or an explicit fma), so host and device agree bit for bit by construction; its error is about one
 * fp64 ulp, i.e. the result is the correctly rounded fp32 value for all but ~1e-8 of the arguments (0 of 4e6 sampled arguments per
 * function differ from (float)libm(double), tests/test_trig.py).  Within one ulp of torch (tests/golden/trig_f32.npz).  Arguments beyond
 * 2^30 return NaN.  About 35 fp64 operations per sine / cosine pair: a quarter of the general-purpose OCML routine it replaces on the device.
 *
 * Compiles as C99 (gcc -mfma -ffp-contract=off) and as HIP device code (-ffp-contract=off): every fused operation is an explicit fma,
 * every other operation a single IEEE operation; `/` is the correctly rounded division on both sides.
 */
#ifndef SIGMA_TRIG_F32_H
#define SIGMA_TRIG_F32_H

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP_DEVICE_COMPILE__)
#define SIGMA_TRIG_FN __host__ __device__ static __forceinline__
#else
#define SIGMA_TRIG_FN static inline
#endif

/* A Horner step whose addend is a literal: fma(a, b, C).  On the device the literal is handed to the instruction in a scalar register pair (two s_mov on the
 * scalar unit) instead of being moved into the destination of a two-address v_fmac_f64 first (two v_mov on the vector unit per step: a quarter of the dynamics
 * phase's vector instructions).  The same IEEE fma either way. */
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __forceinline__ double sigma_fma_kc(double a, double b, double c) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
  return r;
}
#define SIGMA_FMA_KC(a, b, c) sigma_fma_kc((a), (b), (c))
#else
#define SIGMA_FMA_KC(a, b, c) fma((a), (b), (c))
#endif

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
  ps = SIGMA_FMA_KC(ps, z, -2.50507602534068634195e-08);
  ps = SIGMA_FMA_KC(ps, z, 2.75573137070700676789e-06);
  ps = SIGMA_FMA_KC(ps, z, -1.98412698298579493134e-04);
  ps = SIGMA_FMA_KC(ps, z, 8.33333333332248946124e-03);
  ps = SIGMA_FMA_KC(ps, z, -1.66666666666666324348e-01);
  const double s = fma(r * z, ps, r);
  double pc = -1.13596475577881948265e-11;
  pc = SIGMA_FMA_KC(pc, z, 2.08757232129817482790e-09);
  pc = SIGMA_FMA_KC(pc, z, -2.75573143513906633035e-07);
  pc = SIGMA_FMA_KC(pc, z, 2.48015872894767294178e-05);
  pc = SIGMA_FMA_KC(pc, z, -1.38888888888741095749e-03);
  pc = SIGMA_FMA_KC(pc, z, 4.16666666666666019037e-02);
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
  /* numerator and denominator of the reduced argument are SELECTED, then divided once: the same quotient of the same operands as the four branch forms (a
   * vector unit evaluates every branch of an if-chain: four float64 divisions of ~25 instructions each became one) */
  double num = ax, den = 1.0, hi = 0.0, lo = 0.0;
  int red = 1;
  if (ax < 0.4375) { red = 0; }
  else if (ax < 0.6875) { num = 2.0 * ax - 1.0; den = 2.0 + ax; hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17; }
  else if (ax < 1.1875) { num = ax - 1.0; den = ax + 1.0; hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17; }
  else if (ax < 2.4375) { num = ax - 1.5; den = 1.0 + 1.5 * ax; hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17; }
  else { num = -1.0; den = ax; hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17; }
  const double x = red ? num / den : ax;
  const double z = x * x, w = z * z;
  double s1 = 1.62858201153657823623e-02;
  s1 = SIGMA_FMA_KC(s1, w, 4.97687799461593236017e-02);
  s1 = SIGMA_FMA_KC(s1, w, 6.66107313738753120669e-02);
  s1 = SIGMA_FMA_KC(s1, w, 9.09088713343650656196e-02);
  s1 = SIGMA_FMA_KC(s1, w, 1.42857142725034663711e-01);
  s1 = SIGMA_FMA_KC(s1, w, 3.33333333333329318027e-01);
  s1 = s1 * z;
  double s2 = -3.65315727442169155270e-02;
  s2 = SIGMA_FMA_KC(s2, w, -5.83357013379057348645e-02);
  s2 = SIGMA_FMA_KC(s2, w, -7.69187620504482999495e-02);
  s2 = SIGMA_FMA_KC(s2, w, -1.11111104054623557880e-01);
  s2 = SIGMA_FMA_KC(s2, w, -1.99999999998764832476e-01);
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

#endif /* SIGMA_TRIG_F32_H */
