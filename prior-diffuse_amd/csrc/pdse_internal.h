// internal helpers shared by the translation units of libpdse.so (not part of the ABI)
#ifndef PDSE_INTERNAL_H
#define PDSE_INTERNAL_H
#include <hip/hip_runtime.h>

#include "pdse.h"

void pdse_set_error(const char* msg);
// returns 0 when the last launch was accepted, else records "<what>: <hip error>" and returns 1
int pdse_check_launch(const char* what);
int pdse_check_hip(hipError_t e, const char* what);
// hipFuncAttributeMaxDynamicSharedMemorySize, once per (function, device): a plan may be bound to any device of the
// process (pdse_plan_set_device), and the attribute belongs to the device the function is loaded on.  `mask` is the
// caller's per-function record (bit d: done on device d).  Returns 0 on success.
int pdse_lds_attr(const void* fn, unsigned long long* mask, const char* what);
// Diagnostic hooks (clock-stamp traces, stage masks) exist only in -DPDSE_DIAG builds: in the product library the
// environment cannot change a launch (a stage mask gives wrong results; a trace mallocs / synchronises inside a launch,
// which is illegal during hipGraph capture).
#ifdef PDSE_DIAG
#define PDSE_DIAG_ENV(name) (getenv(name))
#else
#define PDSE_DIAG_ENV(name) ((const char*)nullptr)
#endif

int pdse_gconv_launch(const pdse_gconv_desc* d, hipStream_t s);
int pdse_gconv2_launch(const pdse_gconv_desc* d, hipStream_t s);  // korder 1, validated by pdse_gconv_launch
int pdse_gconv3_launch(const pdse_gconv_desc* d, hipStream_t s);  // korder 2 (split-bf16 BIGLU), validated there too
int pdse_gconv4_launch(const pdse_gconv_desc* d, hipStream_t s);  // korder 3 (split-bf16 GEMM-shaped LINEAR / GLU)
int pdse_time_launch(const pdse_time_desc* d, hipStream_t s);
int pdse_ew_launch(const pdse_ew_desc* d, hipStream_t s);
int pdse_compand_launch(const pdse_compand_desc* d, hipStream_t s);
int pdse_wavprep_launch(const pdse_wavprep_desc* d, hipStream_t s);
int pdse_ola_launch(const pdse_ola_desc* d, hipStream_t s);
int pdse_sigma_launch(const pdse_sigma_desc* d, hipStream_t s);
int pdse_ln_launch(const pdse_ln_desc* d, hipStream_t s);
int pdse_lstm_launch(const pdse_lstm_desc* d, hipStream_t s);
int pdse_rowln_launch(const pdse_rowln_desc* d, hipStream_t s);
int pdse_chln_launch(const pdse_chln_desc* d, hipStream_t s);
int pdse_attn_launch(const pdse_attn_desc* d, hipStream_t s);
int pdse_gru_launch(const pdse_gru_desc* d, hipStream_t s);
int pdse_gncomb_launch(const pdse_gncomb_desc* d, hipStream_t s);
int pdse_aham_launch(const pdse_aham_desc* d, hipStream_t s);
int pdse_qsample_launch(const pdse_qsample_desc* d, hipStream_t s);
int pdse_transpose_launch(const pdse_transpose_desc* d, hipStream_t s);
int pdse_tcm_launch(const pdse_tcm_desc* d, hipStream_t s);
int pdse_crm_launch(const pdse_crm_desc* d, hipStream_t s);
int pdse_gcrnlast_launch(const pdse_gcrnlast_desc* d, hipStream_t s);
int pdse_maskloss_launch(const pdse_maskloss_desc* d, hipStream_t s);
int pdse_glstm_launch(const pdse_glstm_desc* d, hipStream_t s);
int pdse_glstmp_launch(const pdse_glstmp_desc* d, hipStream_t s);   /* csrc/lstmp.hip */
int pdse_tcm2_launch(const pdse_tcm2_desc* d, hipStream_t s);
int pdse_tcm2s_launch(const pdse_tcm2s_desc* d, hipStream_t s);   /* csrc/tcm2.hip: the whole stack as one launch */
int pdse_bglu_launch(const pdse_bglu_desc* d, hipStream_t s);
int pdse_planes_launch(const pdse_planes_desc* d, hipStream_t s);
int pdse_dense_launch(const pdse_dense_desc* d, hipStream_t s);     /* csrc/dense.hip */
int pdse_rowlnb_launch(const pdse_rowlnb_desc* d, hipStream_t s);
int pdse_gru3_launch(const pdse_gru_desc* d, hipStream_t s);   /* csrc/gru3.hip, reached through pdse_gru_launch */
#endif
