// The values of every experiment macro of this translation unit as one string (nmod_build_info, include/nanomod_hip.h):
// the phase-skip / variant switches compile into the product kernels, so the shipped binary says what it was built with.
// Include AFTER the kernel headers (their #ifndef defaults must have been seen).
#pragma once
#define NMOD_STR2(x) #x
#define NMOD_STR(x) NMOD_STR2(x)
#ifdef NMOD_EXP
#define NMOD_BI_EXP NMOD_STR(NMOD_EXP)
#else
#define NMOD_BI_EXP "0"
#endif
#ifdef NMOD_SKIP
#define NMOD_BI_SKIP NMOD_STR(NMOD_SKIP)
#else
#define NMOD_BI_SKIP "0"
#endif
#ifdef NMOD_HIST_WAVES
#define NMOD_BI_HIST_WAVES NMOD_STR(NMOD_HIST_WAVES)
#else
#define NMOD_BI_HIST_WAVES "4"
#endif
#ifdef NMOD_WIDE_I16_WORDS
#define NMOD_BI_WIDE_I16_WORDS NMOD_STR(NMOD_WIDE_I16_WORDS)
#else
#define NMOD_BI_WIDE_I16_WORDS "2048"
#endif
#ifdef NMOD_SWZ_MASK
#define NMOD_BI_SWZ_MASK NMOD_STR(NMOD_SWZ_MASK)
#else
#define NMOD_BI_SWZ_MASK "0"
#endif
#ifdef NMOD_PK_SELECT
#define NMOD_BI_PK_SELECT "1"
#else
#define NMOD_BI_PK_SELECT "0"
#endif
#ifdef NMOD_CE_BUILTIN
#define NMOD_BI_CE_BUILTIN "1"
#else
#define NMOD_BI_CE_BUILTIN "0"
#endif
#ifdef NMOD_XOR4_BANKS
#define NMOD_BI_XOR4_BANKS "1"
#else
#define NMOD_BI_XOR4_BANKS "0"
#endif
#ifdef NMOD_NO_GRID
#define NMOD_BI_NO_GRID "1"
#else
#define NMOD_BI_NO_GRID "0"
#endif
#ifdef NMOD_CNT_SKIP
#define NMOD_BI_CNT_SKIP NMOD_STR(NMOD_CNT_SKIP)
#else
#define NMOD_BI_CNT_SKIP "0"
#endif
#ifdef NMOD_CNT_WAVES
#define NMOD_BI_CNT_WAVES NMOD_STR(NMOD_CNT_WAVES)
#else
#define NMOD_BI_CNT_WAVES "4"
#endif
#ifdef NMOD_KS_TOPS
#define NMOD_BI_KS_TOPS NMOD_STR(NMOD_KS_TOPS)
#else
#define NMOD_BI_KS_TOPS "1"
#endif
#ifdef NMOD_WIDE_TOPS
#define NMOD_BI_WIDE_TOPS NMOD_STR(NMOD_WIDE_TOPS)
#else
#define NMOD_BI_WIDE_TOPS "0"
#endif
#ifdef NMOD_CW_OR3
#define NMOD_BI_CW_OR3 NMOD_STR(NMOD_CW_OR3)
#else
#define NMOD_BI_CW_OR3 "1"
#endif
#ifdef NMOD_CNT_TAILS
#define NMOD_BI_CNT_TAILS NMOD_STR(NMOD_CNT_TAILS)
#else
#define NMOD_BI_CNT_TAILS "1"
#endif
#ifdef NMOD_WIDE_TAILS
#define NMOD_BI_WIDE_TAILS NMOD_STR(NMOD_WIDE_TAILS)
#else
#define NMOD_BI_WIDE_TAILS "1"
#endif
#define NMOD_BUILD_FLAGS                                                                                              \
  "NMOD_SKIP=" NMOD_BI_SKIP " NMOD_EXP=" NMOD_BI_EXP " NMOD_HIST_WAVES=" NMOD_BI_HIST_WAVES                            \
  " NMOD_WIDE_I16_WORDS=" NMOD_BI_WIDE_I16_WORDS " NMOD_SWZ_MASK=" NMOD_BI_SWZ_MASK \
  " NMOD_PK_SELECT=" NMOD_BI_PK_SELECT " NMOD_CE_BUILTIN=" NMOD_BI_CE_BUILTIN     \
  " NMOD_XOR4_BANKS=" NMOD_BI_XOR4_BANKS " NMOD_NO_GRID=" NMOD_BI_NO_GRID " NMOD_CNT_SKIP=" NMOD_BI_CNT_SKIP \
  " NMOD_CNT_WAVES=" NMOD_BI_CNT_WAVES " NMOD_KS_TOPS=" NMOD_BI_KS_TOPS " NMOD_WIDE_TOPS=" NMOD_BI_WIDE_TOPS \
  " NMOD_CW_OR3=" NMOD_BI_CW_OR3 " NMOD_CNT_TAILS=" NMOD_BI_CNT_TAILS " NMOD_WIDE_TAILS=" NMOD_BI_WIDE_TAILS
