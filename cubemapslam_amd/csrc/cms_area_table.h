// cms_area_table.h -- the window-unfolding rules of Frame::GetFeaturesInArea (src/Frame.cpp:251-716) as data.
//
// The reference walks a search window [x-r, x+r] x [y-r, y+r] over the 50 x 50 cell grid of the cube face that holds (x, y) and,
// when the window leaves that face, over cells of the neighbouring faces -- 41 hand-written cases, 90 AddCells calls, each with
// its own index arithmetic (some mirrored or transposed, a few that are supersets of the window or name an unexpected face).
// Candidate ORDER is the order of those calls, so the rules are reproduced literally, one table row per AddCells call:
//   face, then four cell bounds (x0, x1, y0, y1), each an expression over ten window-derived cell numbers.
// cms_area_rects() evaluates the rows of the case a query falls into; AddCells' own clamping to [0, 49] is done by the caller.
#ifndef CMS_AREA_TABLE_H
#define CMS_AREA_TABLE_H
#include <stdint.h>

#define CMS_AREA_G 50     /* CUBEFACE_GRID_ROWS == CUBEFACE_GRID_COLS (Frame.h:43-44) */
// The same source compiles for the device and, with CMS_AREA_HOST_EMU, for the host (tests/emu/area_emu.cpp checks the table
// against the oracle's branch-by-branch transcription without a GPU).
#ifdef CMS_AREA_HOST_EMU
#include <math.h>
#define AREA_TAB static const
#define AREA_FN inline
#else
#define AREA_TAB static const __device__
#define AREA_FN __device__ __forceinline__
#endif

// cell numbers derived from the window (all "(int)floor(v * mfGridElementLengthInv)" in float arithmetic):
enum { AV_A, AV_B, AV_C, AV_D,      // xStart, xEnd, yStart, yEnd                      (inside the face)
       AV_BO, AV_DO,                // xEnd - faceW, yEnd - faceH                      (continuation past the right / lower edge)
       AV_AU, AV_CU,                // xStart + faceW, yStart + faceH                  (continuation past the left / upper edge)
       AV_AN, AV_CN,                // -xStart, -yStart                                (mirrored continuation)
       AV_COUNT };
// bound expressions: 0, G-1, G, a value, or G - value - 1
#define AE_Z 0
#define AE_G1 1
#define AE_GG 2
#define AE_V(v) (3 + (v))
#define AE_M(v) (3 + AV_COUNT + (v))

struct CmsAreaRect { int8_t face, x0, x1, y0, y1; };
struct CmsAreaCase { int8_t n; CmsAreaRect r[3]; };

#define AF 0 /* FRONT */
#define AL 1 /* LEFT  */
#define AR 2 /* RIGHT */
#define AU 3 /* UPPER */
#define AW 4 /* LOWER */
#define vA AE_V(AV_A)
#define vB AE_V(AV_B)
#define vC AE_V(AV_C)
#define vD AE_V(AV_D)
#define vBO AE_V(AV_BO)
#define vDO AE_V(AV_DO)
#define vAU AE_V(AV_AU)
#define vCU AE_V(AV_CU)
#define vAN AE_V(AV_AN)
#define vCN AE_V(AV_CN)
#define mA AE_M(AV_A)
#define mB AE_M(AV_B)
#define mC AE_M(AV_C)
#define mD AE_M(AV_D)
#define mBO AE_M(AV_BO)
#define mDO AE_M(AV_DO)
#define mAU AE_M(AV_AU)
#define mCU AE_M(AV_CU)
#define mAN AE_M(AV_AN)
#define mCN AE_M(AV_CN)
#define Z_ AE_Z
#define G1 AE_G1
#define GG AE_GG
#define NONE {0, 0, 0, 0, 0}

// window inside the face in x, leaving it in y: [face][0 = bYOverflow, 1 = underflow]            (Frame.cpp:284-400)
AREA_TAB CmsAreaCase kAreaXin[5][2] = {
  /* FRONT */ {{2, {{AF, vA, vB, vC, G1}, {AW, vA, vB, Z_, vDO}, NONE}}, {2, {{AU, vA, vB, vCU, G1}, {AF, vA, vB, Z_, vD}, NONE}}},
  /* LEFT  */ {{2, {{AL, vA, vB, vC, G1}, {AW, Z_, vDO, mB, mA}, NONE}}, {2, {{AU, Z_, vCN, vA, vB}, {AL, vA, vB, Z_, vD}, NONE}}},
  /* RIGHT */ {{2, {{AR, vA, vB, vC, G1}, {AW, mDO, G1, vA, vB}, NONE}}, {2, {{AU, vCU, G1, mB, mA}, {AR, vA, vB, Z_, vD}, NONE}}},
  /* UPPER */ {{2, {{AU, vA, vB, vC, G1}, {AF, vA, vB, Z_, vDO}, NONE}}, {1, {{AW, vA, vB, Z_, vD}, NONE, NONE}}},
  /* LOWER */ {{1, {{AW, vA, vB, vC, G1}, NONE, NONE}},                   {2, {{AF, vA, vB, vCU, G1}, {AW, vA, vB, Z_, vD}, NONE}}},
};
// window inside the face in y, leaving it in x: [face][0 = bXOverflow, 1 = underflow]            (Frame.cpp:402-497)
AREA_TAB CmsAreaCase kAreaYin[5][2] = {
  /* FRONT */ {{2, {{AF, vA, G1, vC, vD}, {AR, Z_, vBO, vC, vD}, NONE}}, {2, {{AL, vAU, G1, vC, vD}, {AF, Z_, vB, vC, vD}, NONE}}},
  /* LEFT  */ {{2, {{AF, Z_, vBO, vC, vD}, {AL, vA, G1, vC, vD}, NONE}}, {1, {{AL, Z_, vB, vC, vD}, NONE, NONE}}},
  /* RIGHT */ {{1, {{AR, vA, G1, vC, vD}, NONE, NONE}},                   {2, {{AF, vAU, G1, vC, vD}, {AR, Z_, vB, vC, vD}, NONE}}},
  /* UPPER */ {{2, {{AU, vA, G1, vC, vD}, {AR, mD, mC, Z_, vBO}, NONE}}, {2, {{AL, vC, vD, Z_, vAN}, {AU, Z_, vB, vC, vD}, NONE}}},
  /* LOWER */ {{2, {{AW, vA, G1, vC, vD}, {AR, vC, vD, mBO, GG}, NONE}}, {2, {{AL, mD, mC, vAU, G1}, {AW, Z_, vB, vC, vD}, NONE}}},
};
// window leaving the face in x and y: [face][0 = XO&YO, 1 = XU&YO, 2 = XO&YU, 3 = XU&YU]         (Frame.cpp:499-713)
AREA_TAB CmsAreaCase kAreaCorner[5][4] = {
  /* FRONT */ {{3, {{AF, vA, G1, vC, G1}, {AR, Z_, vBO, vC, G1}, {AW, vA, G1, Z_, vDO}}},
               {3, {{AF, Z_, vB, vC, G1}, {AL, vAU, G1, vC, G1}, {AW, Z_, vB, Z_, vDO}}},
               {3, {{AF, vA, G1, Z_, vD}, {AR, Z_, vBO, Z_, vD}, {AU, vA, G1, mCU, G1}}},
               {3, {{AF, Z_, vB, Z_, vD}, {AL, mAU, G1, Z_, vD}, {AU, Z_, vB, mCU, G1}}}},
  /* LEFT  */ {{3, {{AL, vA, G1, vC, G1}, {AF, Z_, vBO, vC, G1}, {AW, Z_, vDO, Z_, mA}}},
               {2, {{AL, Z_, vB, vC, G1}, {AW, Z_, vDO, mB, G1}, NONE}},
               {3, {{AL, vA, G1, Z_, vD}, {AF, Z_, vBO, Z_, vD}, {AU, Z_, vCN, vA, G1}}},
               {2, {{AL, Z_, vB, Z_, vD}, {AU, Z_, vCN, Z_, vD}, NONE}}},
  /* RIGHT */ {{2, {{AR, vA, G1, vC, G1}, {AW, mDO, G1, vA, G1}, NONE}},
               {3, {{AR, Z_, vB, vC, G1}, {AF, mAN, G1, vC, G1}, {AW, mDO, G1, Z_, vB}}},
               {2, {{AR, vA, G1, Z_, vD}, {AU, mCN, G1, Z_, mA}, NONE}},
               {3, {{AR, Z_, vB, Z_, vD}, {AF, mAN, G1, Z_, vD}, {AU, mCN, G1, mB, G1}}}},
  /* UPPER */ {{3, {{AU, vA, G1, vC, G1}, {AR, Z_, mC, Z_, vBO}, {AF, vA, G1, Z_, vDO}}},
               {3, {{AU, Z_, vB, vC, G1}, {AL, vC, G1, Z_, vAN}, {AF, Z_, vB, Z_, vDO}}},
               {2, {{AU, vA, G1, Z_, vD}, {AR, mD, G1, Z_, vD}, NONE}},
               {2, {{AU, Z_, vB, Z_, vD}, {AL, Z_, vD, Z_, vAN}, NONE}}},
  /* LOWER */ {{2, {{AW, vA, G1, vC, G1}, {AR, vA, G1, mBO, G1}, NONE}},
               {2, {{AW, Z_, vB, vC, G1}, {AL, Z_, mC, mAU, G1}, NONE}},
               {3, {{AW, vA, G1, Z_, vD}, {AR, Z_, vD, mBO, GG}, {AF, vA, G1, mCN, G1}}},
               {3, {{AW, Z_, vB, Z_, vD}, {AL, mAN, G1, mAN, GG}, {AF, Z_, vB, mCN, G1}}}},
};
#undef AF
#undef AL
#undef AR
#undef AU
#undef AW
#undef NONE

struct CmsAreaRectI { int face, x0, x1, y0, y1; };

// The rectangles (face + inclusive cell bounds, NOT yet clamped) the reference visits for the window of half size r around the
// canvas position (x, y), in visiting order.  F = cube face size, inv = mfGridElementLengthInv = 150 / (3 F).  Returns their
// number (0: (x, y) on no face).
AREA_FN int cms_area_rects(float x, float y, float r, int F, float inv, CmsAreaRectI out[3]) {
  int face = -1;
  {
    const float i = x / (float)F, j = y / (float)F;          // FaceInCubemap<float>: float quotients (CamModelGeneral.h:458-470)
    if (i >= 0 && i < 1 && j >= 1 && j < 2) face = 1;
    else if (i >= 1 && i < 2 && j >= 0 && j < 1) face = 3;
    else if (i >= 1 && i < 2 && j >= 1 && j < 2) face = 0;
    else if (i >= 1 && i < 2 && j >= 2 && j < 3) face = 4;
    else if (i >= 2 && i < 3 && j >= 1 && j < 2) face = 2;
  }
  if (face < 0) return 0;
  const int cornerX = (int)x / F * F, cornerY = (int)y / F * F;
  const float xin = x - (float)cornerX, yin = y - (float)cornerY;
  const float xs = xin - r, xe = xin + r, ys = yin - r, ye = yin + r;
  const bool xu = xs < 0, xo = xe > (float)(F - 1), yu = ys < 0, yo = ye > (float)(F - 1);
  const bool xinf = !xo && !xu, yinf = !yo && !yu;
  const float Ff = (float)F;
  int v[AV_COUNT];
  v[AV_A] = (int)floorf(xs * inv); v[AV_B] = (int)floorf(xe * inv); v[AV_C] = (int)floorf(ys * inv); v[AV_D] = (int)floorf(ye * inv);
  v[AV_BO] = (int)floorf((xe - Ff) * inv); v[AV_DO] = (int)floorf((ye - Ff) * inv);
  v[AV_AU] = (int)floorf((xs + Ff) * inv); v[AV_CU] = (int)floorf((ys + Ff) * inv);
  v[AV_AN] = (int)floorf((-xs) * inv); v[AV_CN] = (int)floorf((-ys) * inv);
  const CmsAreaCase* cs;
  CmsAreaCase inside;
  if (xinf && yinf) {
    inside.n = 1;
    inside.r[0].face = (int8_t)face; inside.r[0].x0 = vA; inside.r[0].x1 = vB; inside.r[0].y0 = vC; inside.r[0].y1 = vD;
    cs = &inside;
  } else if (xinf) {
    cs = &kAreaXin[face][yo ? 0 : 1];
  } else if (yinf) {
    cs = &kAreaYin[face][xo ? 0 : 1];
  } else {
    int k;                                                    // the reference's else-if chain (Frame.cpp:506-543 etc.)
    if (xo && yo) k = 0; else if (xu && yo) k = 1; else if (xo && yu) k = 2; else k = 3;
    cs = &kAreaCorner[face][k];
  }
  auto ev = [&](int e) -> int {
    if (e == AE_Z) return 0;
    if (e == AE_G1) return CMS_AREA_G - 1;
    if (e == AE_GG) return CMS_AREA_G;
    if (e < 3 + AV_COUNT) return v[e - 3];
    return CMS_AREA_G - v[e - 3 - AV_COUNT] - 1;
  };
  for (int k = 0; k < cs->n; ++k) {
    out[k].face = cs->r[k].face; out[k].x0 = ev(cs->r[k].x0); out[k].x1 = ev(cs->r[k].x1); out[k].y0 = ev(cs->r[k].y0); out[k].y1 = ev(cs->r[k].y1);
  }
  return cs->n;
}
#undef vA
#undef vB
#undef vC
#undef vD
#undef vBO
#undef vDO
#undef vAU
#undef vCU
#undef vAN
#undef vCN
#undef mA
#undef mB
#undef mC
#undef mD
#undef mBO
#undef mDO
#undef mAU
#undef mCU
#undef mAN
#undef mCN
#undef Z_
#undef G1
#undef GG
#endif
