// SPDX-License-Identifier: Apache-2.0
// Interface-compatible with IBM/optimized-number-theoretic-transform-implementations (Copyright IBM Inc., Apache-2.0):
// macro names and meanings follow include/internal/defs.h of that repository so that its unchanged test and benchmark sources compile against
// this directory.  Re-written for this library (bodies, ordering and comments are new); see NOTICE.
/*
 * internal/defs.h -- shared macros of the drop-in boundary.
 *
 * Name-compatible with reference include/internal/defs.h:10-94 so that the
 * reference's tests/test_correctness.c, tests/bench.c and tests/test_cases.h
 * compile unchanged against this include directory (they use EXTERNC_*, SUCCESS,
 * ERROR, GUARD, GUARD_MSG, UNUSED, WORD_SIZE, HIGH_WORD/LOW_WORD, the
 * HAS_AN_*_POWER predicates, LOOP_UNROLL_* and ALIGN).  Written for this
 * library; only the names and meanings are shared.
 */
#ifndef NTT_MI355X_INTERNAL_DEFS_H
#define NTT_MI355X_INTERNAL_DEFS_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

/* ---- C / C++ linkage brackets ---- */
#ifdef __cplusplus
#  define EXTERNC       extern "C"
#  define EXTERNC_BEGIN extern "C" {
#  define EXTERNC_END   }
#else
#  define EXTERNC
#  define EXTERNC_BEGIN
#  define EXTERNC_END
#endif

/* ---- harness status convention (reference defs.h:20-36) ---- */
#define SUCCESS 0
#define ERROR   (-1)

#define GUARD(call)                          \
  {                                          \
    if((call) != SUCCESS) { return ERROR; }  \
  }

#define GUARD_MSG(call, text) \
  {                           \
    if((call) != SUCCESS) {   \
      printf(text);           \
      return ERROR;           \
    }                         \
  }

#if defined(__GNUC__) || defined(__clang__)
#  define UNUSED   __attribute__((unused))
#  define ALIGN(n) __attribute__((aligned(n)))
#else
#  define UNUSED
#  define ALIGN(n)
#endif

/* ---- machine word of the Shoup precomputation (reference defs.h:44-63) ---- */
#define WORD_SIZE             64UL
#define VMSL_WORD_SIZE        56UL /* s390x path of the reference; not built here */
#define AVX512_IFMA_WORD_SIZE 52UL /* x86 IFMA path of the reference; not built here */

#define WORD_SIZE_MASK (~0UL)
#define HIGH_WORD(x)   ((x) >> WORD_SIZE)
#define LOW_WORD(x)    ((x)&WORD_SIZE_MASK)

#define VMSL_WORD_SIZE_MASK ((1UL << VMSL_WORD_SIZE) - 1)
#define HIGH_VMSL_WORD(x)   (uint64_t)((__uint128_t)(x) >> VMSL_WORD_SIZE)
#define LOW_VMSL_WORD(x)    ((x)&VMSL_WORD_SIZE_MASK)

#define AVX512_IFMA_WORD_SIZE_MASK   ((1UL << AVX512_IFMA_WORD_SIZE) - 1)
#define AVX512_IFMA_MAX_MODULUS      49UL
#define AVX512_IFMA_MAX_MODULUS_MASK (~((1UL << AVX512_IFMA_MAX_MODULUS) - 1))

/* ---- N = 2^m: classify m by masking the single set bit (defs.h:65-74) ---- */
#define ODD_POWER_MASK  0xaaaaaaaaaaaaaaaaUL /* m odd        */
#define REM1_POWER_MASK 0x2222222222222222UL /* m = 1 mod 4  */
#define REM2_POWER_MASK 0x4444444444444444UL /* m = 2 mod 4  */
#define REM3_POWER_MASK 0x8888888888888888UL /* m = 3 mod 4  */

#define HAS_AN_EVEN_POWER(n) (((n)&ODD_POWER_MASK) == 0)
#define HAS_AN_REM1_POWER(n) ((n)&REM1_POWER_MASK)
#define HAS_AN_REM2_POWER(n) ((n)&REM2_POWER_MASK)
#define HAS_AN_REM3_POWER(n) ((n)&REM3_POWER_MASK)

/* ---- loop unrolling hints (defs.h:76-92) ---- */
#if defined(__clang__)
#  define LOOP_UNROLL_2 _Pragma("clang loop unroll_count(2)")
#  define LOOP_UNROLL_4 _Pragma("clang loop unroll_count(4)")
#  define LOOP_UNROLL_8 _Pragma("clang loop unroll_count(8)")
#elif defined(__GNUC__) && (__GNUC__ >= 8)
#  define LOOP_UNROLL_2 _Pragma("GCC unroll 2")
#  define LOOP_UNROLL_4 _Pragma("GCC unroll 4")
#  define LOOP_UNROLL_8 _Pragma("GCC unroll 8")
#else
#  define LOOP_UNROLL_2
#  define LOOP_UNROLL_4
#  define LOOP_UNROLL_8
#endif

/* symbols of libntt_mi355x.so that replace the reference's objects */
#if defined(__GNUC__) || defined(__clang__)
#  define NTT_EXPORT __attribute__((visibility("default")))
#else
#  define NTT_EXPORT
#endif

#endif /* NTT_MI355X_INTERNAL_DEFS_H */
