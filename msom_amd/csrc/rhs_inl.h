// rhs_inl.h -- device helpers shared by the PV-tendency kernels (kernels_fused.hip, kernels_lpw.hip).
#ifndef MSOM_RHS_INL_H
#define MSOM_RHS_INL_H

#include "kernels.h"

// validation build: true divisions in the reference's places; product build: reciprocal multiplies
#ifdef MSOM_STRICT
#define DIVC(x, c, rc) ((x) / (c))
#else
#define DIVC(x, c, rc) ((x) * (rc))
#endif

// -J(p,q), msqg/qg.h:252-262, from 3x3 register windows [dy+1][dx+1]
__device__ __forceinline__ double mjac9(const double (&p)[3][3], const double (&q)[3][3], double D12, double rD12) {
#define P(a, b) p[(b) + 1][(a) + 1]
#define Q(a, b) q[(b) + 1][(a) + 1]
#ifdef MSOM_STRICT
  const double s = (Q(1, 0) - Q(-1, 0)) * (P(0, 1) - P(0, -1)) + (Q(0, -1) - Q(0, 1)) * (P(1, 0) - P(-1, 0)) +
                   Q(1, 0) * (P(1, 1) - P(1, -1)) - Q(-1, 0) * (P(-1, 1) - P(-1, -1)) - Q(0, 1) * (P(1, 1) - P(-1, 1)) +
                   Q(0, -1) * (P(1, -1) - P(-1, -1)) + P(0, 1) * (Q(1, 1) - Q(-1, 1)) - P(0, -1) * (Q(1, -1) - Q(-1, -1)) -
                   P(1, 0) * (Q(1, 1) - Q(1, -1)) + P(-1, 0) * (Q(-1, 1) - Q(-1, -1));
#else
  // product build: the same sum as ONE explicit chain of fused multiply-adds (the terms in the order above), so that every
  // instantiation that inlines it rounds alike whatever the compiler's contraction heuristics see around it
  double s = (Q(1, 0) - Q(-1, 0)) * (P(0, 1) - P(0, -1));
  s = fma(Q(0, -1) - Q(0, 1), P(1, 0) - P(-1, 0), s);
  s = fma(Q(1, 0), P(1, 1) - P(1, -1), s);
  s = fma(-Q(-1, 0), P(-1, 1) - P(-1, -1), s);
  s = fma(-Q(0, 1), P(1, 1) - P(-1, 1), s);
  s = fma(Q(0, -1), P(1, -1) - P(-1, -1), s);
  s = fma(P(0, 1), Q(1, 1) - Q(-1, 1), s);
  s = fma(-P(0, -1), Q(1, -1) - Q(-1, -1), s);
  s = fma(-P(1, 0), Q(1, 1) - Q(1, -1), s);
  s = fma(P(-1, 0), Q(-1, 1) - Q(-1, -1), s);
#endif
#undef P
#undef Q
  return DIVC(s, D12, rD12);
}

// whole-wavefront DPP shifts of a double (two 32-bit moves, no LDS crossbar): value held by lane - 1 / lane + 1;
// lane 0 / lane 63 receive 0
__device__ __forceinline__ double lane_below(double v) {  // wave_shr:1
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x138, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x138, 0xf, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double lane_above(double v) {  // wave_shl:1
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x130, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x130, 0xf, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

#endif
