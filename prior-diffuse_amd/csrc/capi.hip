// capi.hip — the extern "C" surface of libpdse.so (include/pdse.h): argument checks,
// per-thread error text, direct launches and recorded plans (replayable, graph-capturable).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>

#include "pdse.h"
#include "pdse_internal.h"

static thread_local std::string g_err;

void pdse_set_error(const char* msg) { g_err = msg ? msg : "unknown error"; }

int pdse_check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return 1;
}

int pdse_check_launch(const char* what) { return pdse_check_hip(hipGetLastError(), what); }

int pdse_lds_attr(const void* fn, unsigned long long* mask, const char* what) {
  int dev = 0;
  if (pdse_check_hip(hipGetDevice(&dev), what)) return 1;
  if (dev >= 0 && dev < 64 && ((*mask >> dev) & 1ull)) return 0;
  if (pdse_check_hip(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), what)) return 1;
  if (dev >= 0 && dev < 64) *mask |= 1ull << dev;
  return 0;
}

union pdse_any_desc {
  pdse_gconv_desc gconv;
  pdse_time_desc time;
  pdse_ew_desc ew;
  pdse_compand_desc compand;
  pdse_wavprep_desc wavprep;
  pdse_ola_desc ola;
  pdse_sigma_desc sigma;
  pdse_ln_desc ln;
  pdse_lstm_desc lstm;
  pdse_rowln_desc rowln;
  pdse_chln_desc chln;
  pdse_attn_desc attn;
  pdse_gru_desc gru;
  pdse_gncomb_desc gncomb;
  pdse_aham_desc aham;
  pdse_qsample_desc qsample;
  pdse_transpose_desc transpose;
  pdse_tcm_desc tcm;
  pdse_tcm2_desc tcm2;
  pdse_tcm2s_desc tcm2s;
  pdse_dense_desc dense;
  pdse_rowlnb_desc rowlnb;
  pdse_crm_desc crm;
  pdse_gcrnlast_desc gcrnlast;
  pdse_maskloss_desc maskloss;
  pdse_glstm_desc glstm;
  pdse_glstmp_desc glstmp;
  pdse_bglu_desc bglu;
  pdse_planes_desc planes;
};

struct pdse_op {
  int kind;
  int tag;
  pdse_any_desc d;
};

// a device allocation owned by a plan that was loaded from a file (pdse_plan_load)
struct pdse_region {
  std::string name;      // empty: internal; else the name an input / output is found by (pdse_plan_region)
  void* ptr = nullptr;
  uint64_t bytes = 0;
};

struct pdse_plan {
  std::vector<pdse_op> ops;
  std::vector<pdse_region> regions;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  int device = -1;   // -1: whatever device is current on the calling thread
};

// Makes the plan's device current for the duration of a call and restores the caller's afterwards: every buffer of a
// plan lives on one device, and a stream handle of device N used while device 0 is current fails (or worse, launches on
// the wrong device).  One process per GPU with LOCAL_RANK != 0 is the case that needs it.
struct device_guard {
  int prev = -1;
  bool ok = true;
  explicit device_guard(const pdse_plan* p) {
    if (!p || p->device < 0) return;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != p->device) ok = !pdse_check_hip(hipSetDevice(p->device), "set device");
    else prev = -1;
  }
  ~device_guard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

static int op_size(int kind) {
  switch (kind) {
    case PDSE_OP_GCONV: return (int)sizeof(pdse_gconv_desc);
    case PDSE_OP_TIME: return (int)sizeof(pdse_time_desc);
    case PDSE_OP_EW: return (int)sizeof(pdse_ew_desc);
    case PDSE_OP_COMPAND: return (int)sizeof(pdse_compand_desc);
    case PDSE_OP_WAVPREP: return (int)sizeof(pdse_wavprep_desc);
    case PDSE_OP_OLA: return (int)sizeof(pdse_ola_desc);
    case PDSE_OP_SIGMA: return (int)sizeof(pdse_sigma_desc);
    case PDSE_OP_LN: return (int)sizeof(pdse_ln_desc);
    case PDSE_OP_LSTM: return (int)sizeof(pdse_lstm_desc);
    case PDSE_OP_ROWLN: return (int)sizeof(pdse_rowln_desc);
    case PDSE_OP_CHLN: return (int)sizeof(pdse_chln_desc);
    case PDSE_OP_ATTN: return (int)sizeof(pdse_attn_desc);
    case PDSE_OP_GRU: return (int)sizeof(pdse_gru_desc);
    case PDSE_OP_GNCOMB: return (int)sizeof(pdse_gncomb_desc);
    case PDSE_OP_AHAM: return (int)sizeof(pdse_aham_desc);
    case PDSE_OP_QSAMPLE: return (int)sizeof(pdse_qsample_desc);
    case PDSE_OP_TRANSPOSE: return (int)sizeof(pdse_transpose_desc);
    case PDSE_OP_TCM: return (int)sizeof(pdse_tcm_desc);
    case PDSE_OP_TCM2: return (int)sizeof(pdse_tcm2_desc);
    case PDSE_OP_TCM2S: return (int)sizeof(pdse_tcm2s_desc);
    case PDSE_OP_DENSE: return (int)sizeof(pdse_dense_desc);
    case PDSE_OP_ROWLNB: return (int)sizeof(pdse_rowlnb_desc);
    case PDSE_OP_CRM: return (int)sizeof(pdse_crm_desc);
    case PDSE_OP_GCRNLAST: return (int)sizeof(pdse_gcrnlast_desc);
    case PDSE_OP_MASKLOSS: return (int)sizeof(pdse_maskloss_desc);
    case PDSE_OP_GLSTM: return (int)sizeof(pdse_glstm_desc);
    case PDSE_OP_GLSTMP: return (int)sizeof(pdse_glstmp_desc);
    case PDSE_OP_BGLU: return (int)sizeof(pdse_bglu_desc);
    case PDSE_OP_PLANES: return (int)sizeof(pdse_planes_desc);
    default: return -1;
  }
}

static int launch_op(const pdse_op& op, hipStream_t s) {
  switch (op.kind) {
    case PDSE_OP_GCONV: return pdse_gconv_launch(&op.d.gconv, s);
    case PDSE_OP_TIME: return pdse_time_launch(&op.d.time, s);
    case PDSE_OP_EW: return pdse_ew_launch(&op.d.ew, s);
    case PDSE_OP_COMPAND: return pdse_compand_launch(&op.d.compand, s);
    case PDSE_OP_WAVPREP: return pdse_wavprep_launch(&op.d.wavprep, s);
    case PDSE_OP_OLA: return pdse_ola_launch(&op.d.ola, s);
    case PDSE_OP_SIGMA: return pdse_sigma_launch(&op.d.sigma, s);
    case PDSE_OP_LN: return pdse_ln_launch(&op.d.ln, s);
    case PDSE_OP_LSTM: return pdse_lstm_launch(&op.d.lstm, s);
    case PDSE_OP_ROWLN: return pdse_rowln_launch(&op.d.rowln, s);
    case PDSE_OP_CHLN: return pdse_chln_launch(&op.d.chln, s);
    case PDSE_OP_ATTN: return pdse_attn_launch(&op.d.attn, s);
    case PDSE_OP_GRU: return pdse_gru_launch(&op.d.gru, s);
    case PDSE_OP_GNCOMB: return pdse_gncomb_launch(&op.d.gncomb, s);
    case PDSE_OP_AHAM: return pdse_aham_launch(&op.d.aham, s);
    case PDSE_OP_QSAMPLE: return pdse_qsample_launch(&op.d.qsample, s);
    case PDSE_OP_TRANSPOSE: return pdse_transpose_launch(&op.d.transpose, s);
    case PDSE_OP_TCM: return pdse_tcm_launch(&op.d.tcm, s);
    case PDSE_OP_CRM: return pdse_crm_launch(&op.d.crm, s);
    case PDSE_OP_GCRNLAST: return pdse_gcrnlast_launch(&op.d.gcrnlast, s);
    case PDSE_OP_MASKLOSS: return pdse_maskloss_launch(&op.d.maskloss, s);
    case PDSE_OP_GLSTM: return pdse_glstm_launch(&op.d.glstm, s);
    case PDSE_OP_GLSTMP: return pdse_glstmp_launch(&op.d.glstmp, s);
    case PDSE_OP_TCM2: return pdse_tcm2_launch(&op.d.tcm2, s);
    case PDSE_OP_TCM2S: return pdse_tcm2s_launch(&op.d.tcm2s, s);
    case PDSE_OP_DENSE: return pdse_dense_launch(&op.d.dense, s);
    case PDSE_OP_ROWLNB: return pdse_rowlnb_launch(&op.d.rowlnb, s);
    case PDSE_OP_BGLU: return pdse_bglu_launch(&op.d.bglu, s);
    case PDSE_OP_PLANES: return pdse_planes_launch(&op.d.planes, s);
    default: pdse_set_error("plan: unknown op kind"); return 1;
  }
}

extern "C" {

int pdse_abi_version(void) { return PDSE_ABI_VERSION; }
const char* pdse_last_error(void) { return g_err.c_str(); }
int pdse_desc_size(int op_kind) { return op_size(op_kind); }

int pdse_gconv_f32(const pdse_gconv_desc* d, pdse_stream_t s) { return pdse_gconv_launch(d, (hipStream_t)s); }
int pdse_time_embed_f32(const pdse_time_desc* d, pdse_stream_t s) { return pdse_time_launch(d, (hipStream_t)s); }
int pdse_ew_f32(const pdse_ew_desc* d, pdse_stream_t s) { return pdse_ew_launch(d, (hipStream_t)s); }
int pdse_compand_f32(const pdse_compand_desc* d, pdse_stream_t s) { return pdse_compand_launch(d, (hipStream_t)s); }
int pdse_wavprep_f32(const pdse_wavprep_desc* d, pdse_stream_t s) { return pdse_wavprep_launch(d, (hipStream_t)s); }
int pdse_ola_f32(const pdse_ola_desc* d, pdse_stream_t s) { return pdse_ola_launch(d, (hipStream_t)s); }
int pdse_sigma_mask_f32(const pdse_sigma_desc* d, pdse_stream_t s) { return pdse_sigma_launch(d, (hipStream_t)s); }
int pdse_layernorm_f32(const pdse_ln_desc* d, pdse_stream_t s) { return pdse_ln_launch(d, (hipStream_t)s); }
int pdse_lstm_f32(const pdse_lstm_desc* d, pdse_stream_t s) { return pdse_lstm_launch(d, (hipStream_t)s); }
int pdse_rowln_prelu_f32(const pdse_rowln_desc* d, pdse_stream_t s) { return pdse_rowln_launch(d, (hipStream_t)s); }
int pdse_chln_f32(const pdse_chln_desc* d, pdse_stream_t s) { return pdse_chln_launch(d, (hipStream_t)s); }
int pdse_attention_f32(const pdse_attn_desc* d, pdse_stream_t s) { return pdse_attn_launch(d, (hipStream_t)s); }
int pdse_bigru_f32(const pdse_gru_desc* d, pdse_stream_t s) { return pdse_gru_launch(d, (hipStream_t)s); }
int pdse_gn_combine_f32(const pdse_gncomb_desc* d, pdse_stream_t s) { return pdse_gncomb_launch(d, (hipStream_t)s); }
int pdse_aham_f32(const pdse_aham_desc* d, pdse_stream_t s) { return pdse_aham_launch(d, (hipStream_t)s); }
int pdse_qsample_f32(const pdse_qsample_desc* d, pdse_stream_t s) { return pdse_qsample_launch(d, (hipStream_t)s); }
int pdse_transpose_f32(const pdse_transpose_desc* d, pdse_stream_t s) { return pdse_transpose_launch(d, (hipStream_t)s); }
int pdse_tcm_f32(const pdse_tcm_desc* d, pdse_stream_t s) { return pdse_tcm_launch(d, (hipStream_t)s); }
int pdse_crm_f32(const pdse_crm_desc* d, pdse_stream_t s) { return pdse_crm_launch(d, (hipStream_t)s); }
int pdse_gcrnlast_f32(const pdse_gcrnlast_desc* d, pdse_stream_t s) { return pdse_gcrnlast_launch(d, (hipStream_t)s); }
int pdse_masked_mse_f32(const pdse_maskloss_desc* d, pdse_stream_t s) { return pdse_maskloss_launch(d, (hipStream_t)s); }
int pdse_glstm_f32(const pdse_glstm_desc* d, pdse_stream_t s) { return pdse_glstm_launch(d, (hipStream_t)s); }
int pdse_glstm_persistent_f32(const pdse_glstmp_desc* d, pdse_stream_t s) { return pdse_glstmp_launch(d, (hipStream_t)s); }
int pdse_tcm2_bf16x3(const pdse_tcm2_desc* d, pdse_stream_t s) { return pdse_tcm2_launch(d, (hipStream_t)s); }
int pdse_tcm2_stack_bf16x3(const pdse_tcm2s_desc* d, pdse_stream_t s) { return pdse_tcm2s_launch(d, (hipStream_t)s); }
int pdse_dense_layer_bf16x3(const pdse_dense_desc* d, pdse_stream_t s) { return pdse_dense_launch(d, (hipStream_t)s); }
int pdse_rowln_blocked_f32(const pdse_rowlnb_desc* d, pdse_stream_t s) { return pdse_rowlnb_launch(d, (hipStream_t)s); }
int pdse_bglu_planes(const pdse_bglu_desc* d, pdse_stream_t s) { return pdse_bglu_launch(d, (hipStream_t)s); }
int pdse_split_planes(const pdse_planes_desc* d, pdse_stream_t s) { return pdse_planes_launch(d, (hipStream_t)s); }

int pdse_plan_create(pdse_plan** out) {
  if (!out) {
    pdse_set_error("plan_create: null out");
    return 1;
  }
  *out = new (std::nothrow) pdse_plan();
  if (!*out) {
    pdse_set_error("plan_create: out of memory");
    return 1;
  }
  return 0;
}

int pdse_plan_add(pdse_plan* p, int op_kind, const void* desc, int tag) {
  const int sz = op_size(op_kind);
  if (!p || !desc || sz < 0) {
    pdse_set_error("plan_add: bad argument");
    return 1;
  }
  if (p->exec) {
    pdse_set_error("plan_add: plan already captured into a graph");
    return 1;
  }
  pdse_op op;
  op.kind = op_kind;
  op.tag = tag;
  memcpy(&op.d, desc, (size_t)sz);
  p->ops.push_back(op);
  return 0;
}

int pdse_plan_size(const pdse_plan* p) { return p ? (int)p->ops.size() : -1; }

int pdse_plan_set_device(pdse_plan* p, int device) {
  int n = 0;
  if (!p || device < -1 || (device >= 0 && (hipGetDeviceCount(&n) != hipSuccess || device >= n))) {
    pdse_set_error("plan_set_device: no such device");
    return 1;
  }
  if (p->exec) {
    pdse_set_error("plan_set_device: plan already captured into a graph");
    return 1;
  }
  p->device = device;
  return 0;
}

int pdse_plan_clear(pdse_plan* p) {
  if (!p) {
    pdse_set_error("plan_clear: null plan");
    return 1;
  }
  if (p->exec) (void)hipGraphExecDestroy(p->exec);
  if (p->graph) (void)hipGraphDestroy(p->graph);
  p->exec = nullptr;
  p->graph = nullptr;
  p->ops.clear();
  return 0;
}

int pdse_plan_run_range(pdse_plan* p, int begin, int end, pdse_stream_t s) {
  if (!p || begin < 0 || end > (int)p->ops.size() || begin > end) {
    pdse_set_error("plan_run_range: bad range");
    return 1;
  }
  device_guard dg(p);
  if (!dg.ok) return 1;
  for (int i = begin; i < end; ++i)
    if (int rc = launch_op(p->ops[i], (hipStream_t)s)) return rc;
  return 0;
}

int pdse_plan_run(pdse_plan* p, pdse_stream_t s) { return pdse_plan_run_range(p, 0, p ? (int)p->ops.size() : 0, s); }

int pdse_plan_build_graph(pdse_plan* p, pdse_stream_t s) {
  if (!p || !s) {
    pdse_set_error("plan_build_graph: needs a plan and a non-default stream");
    return 1;
  }
  if (p->exec) return 0;
  device_guard dg(p);
  if (!dg.ok) return 1;
  hipStream_t st = (hipStream_t)s;
  if (pdse_check_hip(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal), "begin capture")) return 1;
  int rc = pdse_plan_run(p, s);
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(st, &g);
  if (rc) {
    if (g) (void)hipGraphDestroy(g);
    return rc;
  }
  if (pdse_check_hip(e, "end capture")) return 1;
  hipGraphExec_t ex = nullptr;
  if (pdse_check_hip(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0), "graph instantiate")) {
    (void)hipGraphDestroy(g);
    return 1;
  }
  p->graph = g;
  p->exec = ex;
  return 0;
}

int pdse_plan_launch_graph(pdse_plan* p, pdse_stream_t s) {
  if (!p || !p->exec) {
    pdse_set_error("plan_launch_graph: no captured graph");
    return 1;
  }
  device_guard dg(p);
  if (!dg.ok) return 1;
  return pdse_check_hip(hipGraphLaunch(p->exec, (hipStream_t)s), "graph launch");
}

int pdse_plan_time_ops(pdse_plan* p, int begin, int end, pdse_stream_t s, float* ms_out) {
  if (!p || !ms_out || begin < 0 || end > (int)p->ops.size() || begin > end) {
    pdse_set_error("plan_time_ops: bad argument");
    return 1;
  }
  device_guard dg(p);
  if (!dg.ok) return 1;
  hipStream_t st = (hipStream_t)s;
  const int n = end - begin;
  std::vector<hipEvent_t> ev((size_t)n + 1);
  for (auto& e : ev)
    if (pdse_check_hip(hipEventCreate(&e), "event create")) return 1;
  int rc = 0;
  (void)hipEventRecord(ev[0], st);
  for (int i = 0; i < n && !rc; ++i) {
    rc = launch_op(p->ops[begin + i], st);
    (void)hipEventRecord(ev[i + 1], st);
  }
  if (!rc) rc = pdse_check_hip(hipEventSynchronize(ev[n]), "event sync");
  for (int i = 0; i < n && !rc; ++i) rc = pdse_check_hip(hipEventElapsedTime(&ms_out[i], ev[i], ev[i + 1]), "elapsed");
  for (auto& e : ev) (void)hipEventDestroy(e);
  return rc;
}

int pdse_plan_time_tag(pdse_plan* p, int tag, pdse_stream_t s, float* ms_out, int* count_out) {
  if (!p || !ms_out || !count_out) {
    pdse_set_error("plan_time_tag: bad argument");
    return 1;
  }
  device_guard dg(p);
  if (!dg.ok) return 1;
  hipStream_t st = (hipStream_t)s;
  std::vector<hipEvent_t> ev;
  int rc = 0, cnt = 0;
  for (size_t i = 0; i < p->ops.size() && !rc; ++i) {
    const bool hit = p->ops[i].tag == tag;
    if (hit) {
      hipEvent_t a, b;
      if (pdse_check_hip(hipEventCreate(&a), "event create") || pdse_check_hip(hipEventCreate(&b), "event create")) {
        rc = 1;
        break;
      }
      (void)hipEventRecord(a, st);
      rc = launch_op(p->ops[i], st);
      (void)hipEventRecord(b, st);
      ev.push_back(a);
      ev.push_back(b);
      ++cnt;
    } else {
      rc = launch_op(p->ops[i], st);
    }
  }
  if (!rc) rc = pdse_check_hip(hipStreamSynchronize(st), "stream sync");
  float total = 0.f;
  for (size_t i = 0; i + 1 < ev.size() && !rc; i += 2) {
    float ms = 0.f;
    rc = pdse_check_hip(hipEventElapsedTime(&ms, ev[i], ev[i + 1]), "elapsed");
    total += ms;
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  *ms_out = total;
  *count_out = cnt;
  return rc;
}

void pdse_plan_destroy(pdse_plan* p) {
  if (!p) return;
  if (p->exec) (void)hipGraphExecDestroy(p->exec);
  if (p->graph) (void)hipGraphDestroy(p->graph);
  for (auto& r : p->regions)
    if (r.ptr) (void)hipFree(r.ptr);
  delete p;
}

// ---- plan files: a recorded plan with everything it points at, for hosts without the Python builders ---------------------------
// Layout (little endian, every item padded to 8 bytes; written by prior-diffuse_amd/planfile.py):
//   "PDSEPLN1" | u32 abi | u32 nregions | u32 nops | u32 0
//   region: u64 old_base | u64 bytes | u32 kind (low byte: 0 scratch, 1 zeroed, 2 data follows; bits 8..: element type, for
//           readers that want typed views - ignored here) | u32 name_len | name | [data]
//   op:     i32 kind | i32 tag | u32 desc_size | u32 nptr | u32 offsets[nptr] | descriptor bytes
// Loading allocates every region on the current device (or the plan's), uploads / zeroes it and rebases the pointer fields of
// every descriptor (listed by offset) from the recorded addresses to the new ones.
}   // extern "C"
namespace {
struct reader {
  FILE* f;
  bool ok = true;
  void get(void* dst, size_t n) {
    if (ok && fread(dst, 1, n, f) != n) ok = false;
  }
  void skip_pad(size_t n) {
    const size_t pad = (8 - (n & 7)) & 7;
    char z[8];
    if (pad) get(z, pad);
  }
  template <typename T>
  T val() {
    T v{};
    get(&v, sizeof(T));
    return v;
  }
};
}  // namespace
extern "C" {

int pdse_plan_load(const char* path, pdse_plan** out) {
  if (!path || !out) {
    pdse_set_error("plan_load: null argument");
    return 1;
  }
  FILE* f = fopen(path, "rb");
  if (!f) {
    pdse_set_error("plan_load: cannot open the file");
    return 1;
  }
  reader rd{f};
  pdse_plan* p = new (std::nothrow) pdse_plan();
  auto fail = [&](const char* msg) {
    pdse_set_error(msg);
    fclose(f);
    pdse_plan_destroy(p);
    return 1;
  };
  if (!p) return fail("plan_load: out of memory");
  char magic[8];
  rd.get(magic, 8);
  const uint32_t abi = rd.val<uint32_t>(), nreg = rd.val<uint32_t>(), nops = rd.val<uint32_t>();
  (void)rd.val<uint32_t>();
  if (!rd.ok || memcmp(magic, "PDSEPLN1", 8) != 0) return fail("plan_load: not a plan file");
  if (abi != PDSE_ABI_VERSION) return fail("plan_load: the file was written for another ABI version");
  if (nreg > (1u << 20) || nops > (1u << 22)) return fail("plan_load: implausible header");
  std::vector<uint64_t> old_base(nreg);
  std::vector<char> buf;
  for (uint32_t i = 0; i < nreg; ++i) {
    pdse_region r;
    old_base[i] = rd.val<uint64_t>();
    r.bytes = rd.val<uint64_t>();
    const uint32_t kind = rd.val<uint32_t>() & 0xffu, nlen = rd.val<uint32_t>();
    if (!rd.ok || nlen > 256 || kind > 2 || r.bytes == 0 || r.bytes > (1ull << 40)) return fail("plan_load: bad region record");
    r.name.resize(nlen);
    rd.get(&r.name[0], nlen);
    rd.skip_pad(nlen);
    if (pdse_check_hip(hipMalloc(&r.ptr, r.bytes), "plan_load: hipMalloc")) return fail(pdse_last_error());
    p->regions.push_back(r);   // owned from here on (freed by destroy)
    if (kind == 2) {
      buf.resize(r.bytes);
      rd.get(buf.data(), r.bytes);
      rd.skip_pad(r.bytes);
      if (!rd.ok) return fail("plan_load: truncated region data");
      if (pdse_check_hip(hipMemcpy(r.ptr, buf.data(), r.bytes, hipMemcpyHostToDevice), "plan_load: upload")) return fail(pdse_last_error());
    } else {
      if (pdse_check_hip(hipMemset(r.ptr, 0, r.bytes), "plan_load: memset")) return fail(pdse_last_error());
    }
  }
  std::vector<uint32_t> offs;
  for (uint32_t i = 0; i < nops; ++i) {
    pdse_op op;
    op.kind = rd.val<int32_t>();
    op.tag = rd.val<int32_t>();
    const uint32_t dsz = rd.val<uint32_t>(), nptr = rd.val<uint32_t>();
    const int want = op_size(op.kind);
    if (!rd.ok || want < 0 || (uint32_t)want != dsz || nptr > dsz / 8) return fail("plan_load: bad operator record (descriptor size or kind)");
    offs.resize(nptr);
    rd.get(offs.data(), nptr * sizeof(uint32_t));
    rd.skip_pad(nptr * sizeof(uint32_t));
    memset(&op.d, 0, sizeof(op.d));
    rd.get(&op.d, dsz);
    rd.skip_pad(dsz);
    if (!rd.ok) return fail("plan_load: truncated operator record");
    char* const base = reinterpret_cast<char*>(&op.d);
    for (uint32_t k = 0; k < nptr; ++k) {
      if (offs[k] + 8 > dsz || (offs[k] & 7)) return fail("plan_load: bad pointer offset");
      uint64_t v;
      memcpy(&v, base + offs[k], 8);
      if (!v) continue;
      bool found = false;
      for (uint32_t r = 0; r < nreg && !found; ++r) {
        if (v >= old_base[r] && v < old_base[r] + p->regions[r].bytes) {
          v = reinterpret_cast<uint64_t>(p->regions[r].ptr) + (v - old_base[r]);
          found = true;
        }
      }
      if (!found) return fail("plan_load: a descriptor points outside every recorded region");
      memcpy(base + offs[k], &v, 8);
    }
    p->ops.push_back(op);
  }
  fclose(f);
  *out = p;
  return 0;
}

int pdse_plan_region(const pdse_plan* p, const char* name, void** ptr, uint64_t* nbytes) {
  if (!p || !name || !ptr) {
    pdse_set_error("plan_region: null argument");
    return 1;
  }
  for (const auto& r : p->regions) {
    if (r.name == name) {
      *ptr = r.ptr;
      if (nbytes) *nbytes = r.bytes;
      return 0;
    }
  }
  pdse_set_error("plan_region: no region of that name in this plan (plans recorded in this process keep their buffers in the caller)");
  return 1;
}

// copy inputs into the plan's named regions, run it, copy the named outputs out - all on stream s, device to device
static int plan_call(pdse_plan* p, const int n_in, const char* const* in_names, const void* const* in_ptrs, const int n_out,
                     const char* const* out_names, void* const* out_ptrs, pdse_stream_t s) {
  if (!p) {
    pdse_set_error("forward: null plan");
    return 1;
  }
  device_guard dg(p);
  if (!dg.ok) return 1;
  for (int i = 0; i < n_in; ++i) {
    if (!in_ptrs[i]) continue;   // optional input left as the plan holds it
    void* dst;
    uint64_t n;
    if (pdse_plan_region(p, in_names[i], &dst, &n)) return 1;
    if (pdse_check_hip(hipMemcpyAsync(dst, in_ptrs[i], n, hipMemcpyDeviceToDevice, (hipStream_t)s), "forward: input copy")) return 1;
  }
  if (pdse_plan_run(p, s)) return 1;
  for (int i = 0; i < n_out; ++i) {
    if (!out_ptrs[i]) continue;
    void* src;
    uint64_t n;
    if (pdse_plan_region(p, out_names[i], &src, &n)) return 1;
    if (pdse_check_hip(hipMemcpyAsync(out_ptrs[i], src, n, hipMemcpyDeviceToDevice, (hipStream_t)s), "forward: output copy")) return 1;
  }
  return 0;
}

int pdse_prior_forward(pdse_plan* p, const float* feat, float* x_init, pdse_stream_t s) {
  const char* in[] = {"x"};
  const void* ip[] = {feat};
  const char* on[] = {"out"};
  void* op[] = {x_init};
  return plan_call(p, 1, in, ip, 1, on, op, s);
}

int pdse_eps_forward(pdse_plan* p, const float* x, const float* x_init, const float* t, float* eps, pdse_stream_t s) {
  const char* in[] = {"x", "x_init", "t"};
  const void* ip[] = {x, x_init, t};
  const char* on[] = {"out"};
  void* op[] = {eps};
  return plan_call(p, 3, in, ip, 1, on, op, s);
}

int pdse_enhance(pdse_plan* p, const float* wav, const float* x_T, float* wav_out, float* spec_out, pdse_stream_t s) {
  const char* in[] = {"wav", "x_T"};
  const void* ip[] = {wav, x_T};
  const char* on[] = {"wav_out", "spec"};
  void* op[] = {wav_out, spec_out};
  return plan_call(p, 2, in, ip, 2, on, op, s);
}

}  // extern "C"
