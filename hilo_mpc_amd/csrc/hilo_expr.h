// Postfix expression programs evaluated on the device for any scalar type (double, Jet2).
//
// The reference lets the user write path references and constraint functions as CasADi expressions of the model's
// symbols (`nmpc.stage_constraint.constraint = ...`, hilo_mpc/util/modeling.py:930-940; path references
// modeling.py:252-261).  Here such an expression is compiled on the host (hilo_mpc_amd/expr.py) into a flat postfix
// program that lives with the problem constants in LDS; every lane interprets the same program (all control flow is
// wave-uniform), the operand type carries the derivatives.
//
// Encoding: a program is [len, (op, arg) * len/2]; programs are stored back to back.  Opcodes = HILO_X_* (hilo_hip.h).
#pragma once
#include "hilo_ad.h"

namespace hilo {

constexpr int EXPR_STACK = 8;

enum ExprOp : int {
  X_CONST = 0, X_VARX = 1, X_VARU = 2, X_VARP = 3,
  X_ADD = 10, X_SUB = 11, X_MUL = 12, X_DIV = 13, X_NEG = 14, X_SQ = 15, X_SIN = 16, X_COS = 17, X_EXP = 18,
  X_LOG = 19, X_SQRT = 20, X_POWI = 21,
};

template <int n, class T>
__device__ __forceinline__ T pick(const T* a, int idx) {  // register array, wave-uniform index
  T v = a[0];
#pragma unroll
  for (int i = 1; i < n; ++i)
    if (idx == i) v = a[i];
  return v;
}

template <class T>
__device__ __forceinline__ void put(T* st, int idx, const T& v) {
#pragma unroll
  for (int s = 0; s < EXPR_STACK; ++s)
    if (idx == s) st[s] = v;
}

// skip to program `which` in a block of back-to-back programs
__device__ __forceinline__ const double* expr_program(const double* block, int which) {
  for (int q = 0; q < which; ++q) block += 1 + (int)block[0];
  return block;
}

template <int NX, int NU, class T>
__device__ T expr_eval(const double* prog, const T* x, const T* u, const double* p) {
  const int len = __builtin_amdgcn_readfirstlane((int)prog[0]);
  const double* code = prog + 1;
  T st[EXPR_STACK];
#pragma unroll
  for (int i = 0; i < EXPR_STACK; ++i) st[i] = T(0.0);
  int sp = 0;
  for (int q = 0; q < len; q += 2) {
    const int op = __builtin_amdgcn_readfirstlane((int)code[q]);
    const double a = code[q + 1];
    const int ia = __builtin_amdgcn_readfirstlane((int)a);
    if (op < 10) {
      T v;
      if (op == X_CONST) v = T(a);
      else if (op == X_VARX) v = pick<NX>(x, ia);
      else if (op == X_VARU) v = pick<(NU > 0 ? NU : 1)>(u, ia);
      else v = T(p[ia]);
      put(st, sp, v);
      ++sp;
    } else if (op < 14) {  // binary: c (below) op b (top)
      const T b = pick<EXPR_STACK>(st, sp - 1), c = pick<EXPR_STACK>(st, sp - 2);
      T r;
      if (op == X_ADD) r = c + b;
      else if (op == X_SUB) r = c - b;
      else if (op == X_MUL) r = c * b;
      else r = c / b;
      --sp;
      put(st, sp - 1, r);
    } else {  // unary
      const T b = pick<EXPR_STACK>(st, sp - 1);
      T r;
      switch (op) {
        case X_NEG: r = -1.0 * b; break;
        case X_SQ: r = sq(b); break;
        case X_SIN: r = sin(b); break;
        case X_COS: r = cos(b); break;
        case X_EXP: r = exp(b); break;
        case X_LOG: r = log(b); break;
        case X_SQRT: r = sqrt(b); break;
        default: {  // X_POWI: integer power by repeated multiplication (|n| <= 16), negative through the reciprocal
          const int n = ia < 0 ? -ia : ia;
          r = T(1.0);
          for (int e = 0; e < n; ++e) r = r * b;
          if (ia < 0) r = 1.0 / r;
        }
      }
      put(st, sp - 1, r);
    }
  }
  return st[0];
}

// host-side validation of a block of `count` programs: opcodes, variable indices, stack discipline
inline int expr_check(const double* block, int block_len, int count, int nx, int nu, int np, const char** why) {
  int pos = 0;
  for (int k = 0; k < count; ++k) {
    if (pos >= block_len) { *why = "program block too short"; return 1; }
    const int len = (int)block[pos];
    if (len < 2 || len % 2 || pos + 1 + len > block_len) { *why = "bad program length"; return 1; }
    int sp = 0;
    for (int q = 0; q < len; q += 2) {
      const int op = (int)block[pos + 1 + q];
      const int ia = (int)block[pos + 2 + q];
      if (op == X_CONST) ++sp;
      else if (op == X_VARX) { if (ia < 0 || ia >= nx) { *why = "state index out of range"; return 1; } ++sp; }
      else if (op == X_VARU) { if (ia < 0 || ia >= nu) { *why = "input index out of range"; return 1; } ++sp; }
      else if (op == X_VARP) { if (ia < 0 || ia >= np) { *why = "parameter index out of range"; return 1; } ++sp; }
      else if (op >= X_ADD && op <= X_DIV) { if (sp < 2) { *why = "stack underflow"; return 1; } --sp; }
      else if (op >= X_NEG && op <= X_POWI) {
        if (sp < 1) { *why = "stack underflow"; return 1; }
        if (op == X_POWI && (ia < -16 || ia > 16)) { *why = "integer power out of range [-16, 16]"; return 1; }
      } else { *why = "unknown opcode"; return 1; }
      if (sp > EXPR_STACK) { *why = "expression too deep (stack of 8)"; return 1; }
    }
    if (sp != 1) { *why = "program leaves more than one value"; return 1; }
    pos += 1 + len;
  }
  if (pos != block_len) { *why = "trailing data after the last program"; return 1; }
  return 0;
}

}  // namespace hilo
