// cms_detmath.h -- bit-reproducible scalar math shared by host and gfx950 device code.
//
// The extractor's outputs must match the CPU reference bit for bit (key-point angles feed cvRound'ed tap
// positions, ORBExtractor.cpp:83-96), so everything here is plain IEEE add/mul/div evaluated in a fixed order
// (the library is built with -ffp-contract=off): the same source gives the same bits from g++ and from hipcc.
//   cms_cv_round    : cvRound (round half to even)                       -- OpenCV core, used ORBExtractor.cpp:53,94-96
//   cms_fast_atan2  : cv::fastAtan2 degree polynomial                     -- OpenCV core/mathfuncs, used ORBExtractor.cpp:74
//   cms_cosf/sinf   : float cos/sin of the key-point angle (ORBExtractor.cpp:84), evaluated in double
//                     (Cody-Waite reduction + Taylor to 1e-17) and rounded once to float.
#ifndef CMS_DETMATH_H
#define CMS_DETMATH_H

#if defined(__HIPCC__)
#define CMS_HD __host__ __device__ __forceinline__
#else
#define CMS_HD inline
#endif

CMS_HD int cms_cv_round(float v) { return (int)__builtin_rintf(v); }

CMS_HD float cms_fast_atan2(float y, float x) {
  const float k = (float)(180 / 3.1415926535897932384626433832795);
  const float p1 = 0.9997878412794807f * k, p3 = -0.3258083974640975f * k, p5 = 0.1555786518463281f * k,
              p7 = -0.04432655554792128f * k;
  const float eps = (float)2.2204460492503131e-16;
  const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + eps);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + eps);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// sin and cos of x (|x| < ~1e4) in double, then one rounding to float.
CMS_HD void cms_sincosf(float xf, float* s_out, float* c_out) {
  const double x = (double)xf;
  const double two_over_pi = 0.63661977236758134308;
  const double pio2_hi = 1.57079632673412561417e+00;  // 33 bits of pi/2
  const double pio2_lo = 6.07710050650619224932e-11;  // pi/2 - pio2_hi
  const double kd = __builtin_rint(x * two_over_pi);
  const double r = (x - kd * pio2_hi) - kd * pio2_lo;
  const double r2 = r * r;
  // Taylor, |r| <= pi/4: remainder < 1e-17
  double sp = -1.0 / 1307674368000.0;              // -1/15!
  sp = sp * r2 + 1.0 / 6227020800.0;               // 1/13!
  sp = sp * r2 - 1.0 / 39916800.0;                 // 1/11!
  sp = sp * r2 + 1.0 / 362880.0;                   // 1/9!
  sp = sp * r2 - 1.0 / 5040.0;                     // 1/7!
  sp = sp * r2 + 1.0 / 120.0;                      // 1/5!
  sp = sp * r2 - 1.0 / 6.0;                        // 1/3!
  const double sn = r + r * (r2 * sp);
  double cp = 1.0 / 20922789888000.0;              // 1/16!
  cp = cp * r2 - 1.0 / 87178291200.0;              // 1/14!
  cp = cp * r2 + 1.0 / 479001600.0;                // 1/12!
  cp = cp * r2 - 1.0 / 3628800.0;                  // 1/10!
  cp = cp * r2 + 1.0 / 40320.0;                    // 1/8!
  cp = cp * r2 - 1.0 / 720.0;                      // 1/6!
  cp = cp * r2 + 1.0 / 24.0;                       // 1/4!
  cp = cp * r2 - 0.5;
  const double cs = 1.0 + r2 * cp;
  const int q = ((int)kd) & 3;
  double s, c;
  if (q == 0) { s = sn; c = cs; }
  else if (q == 1) { s = cs; c = -sn; }
  else if (q == 2) { s = -sn; c = -cs; }
  else { s = -cs; c = sn; }
  *s_out = (float)s;
  *c_out = (float)c;
}

#endif
