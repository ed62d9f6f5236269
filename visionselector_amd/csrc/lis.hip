// C-ABI entry points of the LIS inference path (kernels in lis_kernels.h).
#include "lis_kernels.h"

#include <stdlib.h>

namespace vsel {
int check_segments(const vsel_segments* seg, bool need_k) { return check_segments_impl(seg, need_k); }
}  // namespace vsel

using namespace vsel;

// ---- two-half software pipeline ----------------------------------------------------------------------------------
// With >= kPipelineMinSegments segments the batch is split in two halves that run on the caller's stream and on a
// library-owned auxiliary stream (fork / join with events): the small latency-bound kernels of one half (projections,
// finish kernels, radix select) execute underneath the HBM-bound sweeps of the other half.  Results are identical to
// the single-stream order (each half is self-contained, every kernel is deterministic).
constexpr int64_t kPipelineMinSegments = 32;

struct AuxStream {
  hipStream_t stream = nullptr;
  hipEvent_t fork[8], join[8];
  unsigned next = 0;
  bool ok = false;
};

static AuxStream* aux_for_current_device() {
  static AuxStream table[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  AuxStream& a = table[dev];
  if (!a.ok) {
    if (hipStreamCreateWithFlags(&a.stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
    for (int i = 0; i < 8; ++i) {
      if (hipEventCreateWithFlags(&a.fork[i], hipEventDisableTiming) != hipSuccess) return nullptr;
      if (hipEventCreateWithFlags(&a.join[i], hipEventDisableTiming) != hipSuccess) return nullptr;
    }
    a.ok = true;
  }
  return &a;
}

// OFF by default: VSEL_PIPELINE=1 in the environment (or vsel_debug_set_pipeline(1)) turns it on.  Measured +4 % tokens/s
// at B=128 on MI355X; it is opt-in because concurrent half-batch launches make per-kernel durations (rocprofv3, HIP
// events) incomparable with the single-stream roofline numbers bench.py reports.
static int g_pipeline_enabled = [] {
  const char* e = getenv("VSEL_PIPELINE");
  return (e && e[0] == '1') ? 1 : 0;
}();
extern "C" void vsel_debug_set_pipeline(int on) { g_pipeline_enabled = on; }

template <typename T, typename TW>
static int select_half(hipStream_t st, const T* h, const vsel_segments* seg, const vsel_scorer* sc, char* ws, const LisPlan& p,
                       T* out, int64_t* idx, float* scores, bool skip_colsum, const int64_t* l2p = nullptr,
                       const int64_t* p2l = nullptr, const float* col_sums = nullptr) {
  int rc = VSEL_OK;
  if (!skip_colsum && !col_sums) rc = run_colsum<T>(st, h, seg, (int)sc->d, ws, p);
  if (rc) return rc;
  rc = run_proj<TW>(st, seg, sc, ws, p, col_sums);
  if (rc) return rc;
  rc = run_score<T>(st, h, seg, sc, ws, p, scores, p2l);
  if (rc) return rc;
  rc = launch_select(st, scores, seg, idx, nullptr);
  if (rc) return rc;
  return launch_gather<T>(st, h, (int)sc->d, seg, idx, out, l2p);
}

template <typename T, typename TW>
static int lis_select_impl(hipStream_t st, const T* h, const vsel_segments* seg, const vsel_scorer* sc, char* ws, T* out,
                           int64_t* idx, float* scores, const int64_t* l2p = nullptr, const int64_t* p2l = nullptr,
                           const float* col_sums = nullptr) {
  const int64_t S = seg->n_seg, d = sc->d;
  AuxStream* aux = (g_pipeline_enabled && S >= kPipelineMinSegments && !prof_enabled() && !l2p && !col_sums)
                       ? aux_for_current_device() : nullptr;
  if (!aux) {
    const LisPlan p = make_plan(S, seg->rows_per_seg, d, sc->hd);
    return select_half<T, TW>(st, h, seg, sc, ws, p, out, idx, scores, false, l2p, p2l, col_sums);
  }
  // halves A = [0, s0), B = [s0, S).  Uniform segments address rows relative to the half's base pointer; ragged
  // segments keep absolute offsets (seg_rows / seg_out are advanced instead).
  const int64_t s0 = S / 2;
  vsel_segments a = *seg, b = *seg;
  a.n_seg = s0;
  b.n_seg = S - s0;
  const T* hb = h;
  T* outb = out;
  int64_t* idxb = idx;
  float* scb = scores;
  if (seg->seg_rows) {
    b.seg_rows = seg->seg_rows + s0;
    b.seg_out = seg->seg_out + s0;
    a.total_rows = b.total_rows = seg->total_rows;   // informational only in the ragged form
  } else {
    a.total_rows = s0 * seg->rows_per_seg;
    a.total_out = s0 * seg->k;
    b.total_rows = (S - s0) * seg->rows_per_seg;
    b.total_out = (S - s0) * seg->k;
    hb = h + a.total_rows * d;
    outb = out + a.total_out * d;
    idxb = idx + a.total_out;
    scb = scores + a.total_rows;
  }
  const LisPlan pa = make_plan(a.n_seg, seg->rows_per_seg, d, sc->hd);
  const LisPlan pb = make_plan(b.n_seg, seg->rows_per_seg, d, sc->hd);
  char* wsa = ws;
  char* wsb = ws + pa.total;
  const unsigned e = aux->next++ & 7u;
  int rc = run_colsum<T>(st, h, &a, (int)d, wsa, pa);
  if (rc) return rc;
  VSEL_HIP_CHECK(hipEventRecord(aux->fork[e], st));
  VSEL_HIP_CHECK(hipStreamWaitEvent(aux->stream, aux->fork[e], 0));
  rc = select_half<T, TW>(aux->stream, h, &a, sc, wsa, pa, out, idx, scores, true);
  if (rc) return rc;
  VSEL_HIP_CHECK(hipEventRecord(aux->join[e], aux->stream));
  rc = select_half<T, TW>(st, hb, &b, sc, wsb, pb, outb, idxb, scb, false);
  if (rc) return rc;
  VSEL_HIP_CHECK(hipStreamWaitEvent(st, aux->join[e], 0));
  return VSEL_OK;
}

extern "C" size_t vsel_lis_workspace_bytes(const vsel_segments* seg, int64_t d, int64_t hd) {
  if (!seg || seg->n_seg < 1 || d < 1 || hd < 1) return 0;
  const size_t whole = make_plan(seg->n_seg, seg->rows_per_seg, d, hd).total;
  if (seg->n_seg < kPipelineMinSegments) return whole;
  const int64_t s0 = seg->n_seg / 2;
  const size_t halves = make_plan(s0, seg->rows_per_seg, d, hd).total + make_plan(seg->n_seg - s0, seg->rows_per_seg, d, hd).total;
  return std::max(whole, halves);
}

static int lis_common_checks(const void* h, const vsel_segments* seg, const vsel_scorer* sc, vsel_dtype hdtype,
                             void* ws, size_t ws_bytes, bool need_k, LisPlan* plan) {
  if (!h) return fail(VSEL_ERR_INVALID, "h is NULL");
  int st = check_segments(seg, need_k);
  if (st) return st;
  st = check_scorer(sc, hdtype);
  if (st) return st;
  *plan = make_plan(seg->n_seg, seg->rows_per_seg, sc->d, sc->hd);
  if (!ws || ws_bytes < plan->total)
    return fail(VSEL_ERR_WORKSPACE, "workspace %zu B < required %zu B", ws_bytes, plan->total);
  if (((uintptr_t)h | (uintptr_t)ws | (uintptr_t)sc->wq | (uintptr_t)sc->wk) & 15)
    return fail(VSEL_ERR_INVALID, "h / workspace / weights must be 16-byte aligned");
  return VSEL_OK;
}

extern "C" int vsel_lis_scores(void* stream, const void* h, vsel_dtype hdtype, const vsel_segments* seg,
                               const vsel_scorer* sc, void* ws, size_t ws_bytes, float* scores) {
  LisPlan p;
  int st = lis_common_checks(h, seg, sc, hdtype, ws, ws_bytes, false, &p);
  if (st) return st;
  if (!scores) return fail(VSEL_ERR_INVALID, "scores is NULL");
  hipStream_t s = (hipStream_t)stream;
  VSEL_PROF_BEGIN(s);
  if (hdtype == VSEL_BF16) return run_scores_w<bf16_t>(s, (const bf16_t*)h, seg, sc, (char*)ws, p, scores);
  return run_scores_w<float>(s, (const float*)h, seg, sc, (char*)ws, p, scores);
}

extern "C" int vsel_topk_select(void* stream, const float* scores, const vsel_segments* seg, int64_t* idx, float* mask) {
  int st = check_segments(seg, true);
  if (st) return st;
  if (!scores || (!idx && !mask)) return fail(VSEL_ERR_INVALID, "scores / outputs NULL");
  VSEL_PROF_BEGIN(stream);
  return launch_select((hipStream_t)stream, scores, seg, idx, mask);
}

extern "C" int vsel_gather_rows(void* stream, const void* h, vsel_dtype hdtype, int64_t d, const vsel_segments* seg,
                                const int64_t* idx, void* out) {
  int st = check_segments(seg, true);
  if (st) return st;
  if (!h || !idx || !out) return fail(VSEL_ERR_INVALID, "NULL pointer");
  if (((uintptr_t)h | (uintptr_t)out) & 15) return fail(VSEL_ERR_INVALID, "h / out must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  VSEL_PROF_BEGIN(s);
  if (hdtype == VSEL_BF16) {
    if (d % 8) return fail(VSEL_ERR_UNSUPPORTED, "D must be a multiple of 8");
    return launch_gather<bf16_t>(s, (const bf16_t*)h, (int)d, seg, idx, (bf16_t*)out);
  }
  if (hdtype == VSEL_F32) {
    if (d % 4) return fail(VSEL_ERR_UNSUPPORTED, "D must be a multiple of 4");
    return launch_gather<float>(s, (const float*)h, (int)d, seg, idx, (float*)out);
  }
  return fail(VSEL_ERR_INVALID, "bad dtype");
}

extern "C" int vsel_lis_select(void* stream, const void* h, vsel_dtype hdtype, const vsel_segments* seg,
                               const vsel_scorer* sc, void* ws, size_t ws_bytes, void* out, int64_t* idx,
                               float* scores) {
  LisPlan p;
  int st = lis_common_checks(h, seg, sc, hdtype, ws, ws_bytes, true, &p);
  if (st) return st;
  if (!out || !idx || !scores) return fail(VSEL_ERR_INVALID, "out / idx / scores is NULL");
  if ((uintptr_t)out & 15) return fail(VSEL_ERR_INVALID, "out must be 16-byte aligned");
  if (ws_bytes < vsel_lis_workspace_bytes(seg, sc->d, sc->hd))
    return fail(VSEL_ERR_WORKSPACE, "workspace %zu B < required %zu B", ws_bytes, vsel_lis_workspace_bytes(seg, sc->d, sc->hd));
  hipStream_t s = (hipStream_t)stream;
  VSEL_PROF_BEGIN(s);
  if (hdtype == VSEL_BF16) {
    if (sc->wdtype == VSEL_BF16)
      return lis_select_impl<bf16_t, bf16_t>(s, (const bf16_t*)h, seg, sc, (char*)ws, (bf16_t*)out, idx, scores);
    return lis_select_impl<bf16_t, float>(s, (const bf16_t*)h, seg, sc, (char*)ws, (bf16_t*)out, idx, scores);
  }
  if (sc->wdtype == VSEL_BF16)
    return lis_select_impl<float, bf16_t>(s, (const float*)h, seg, sc, (char*)ws, (float*)out, idx, scores);
  return lis_select_impl<float, float>(s, (const float*)h, seg, sc, (char*)ws, (float*)out, idx, scores);
}

// Permuted form: the token rows are stored in a PHYSICAL order (Qwen2.5-VL's window order straight out of the merger) and
// the reference semantics are defined on the LOGICAL order it creates with `hidden_states[reverse_indices, :]`
// (EV/token_compression/selector_model.py:179-181).  Scores and indices are produced in logical order, kept rows are
// gathered straight from the physical tensor, so that un-reorder pass (read + write of N x D) is never materialised.
extern "C" int vsel_lis_select_permuted(void* stream, const void* h_physical, vsel_dtype hdtype, const vsel_segments* seg,
                                        const vsel_scorer* sc, void* ws, size_t ws_bytes, const int64_t* logical_to_physical,
                                        const int64_t* physical_to_logical, void* out, int64_t* idx, float* scores) {
  LisPlan p;
  int st = lis_common_checks(h_physical, seg, sc, hdtype, ws, ws_bytes, true, &p);
  if (st) return st;
  if (!out || !idx || !scores || !logical_to_physical || !physical_to_logical) return fail(VSEL_ERR_INVALID, "NULL pointer");
  if ((uintptr_t)out & 15) return fail(VSEL_ERR_INVALID, "out must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  VSEL_PROF_BEGIN(s);
  const int64_t* l2p = logical_to_physical;
  const int64_t* p2l = physical_to_logical;
  if (hdtype == VSEL_BF16) {
    if (sc->wdtype == VSEL_BF16)
      return lis_select_impl<bf16_t, bf16_t>(s, (const bf16_t*)h_physical, seg, sc, (char*)ws, (bf16_t*)out, idx, scores, l2p, p2l);
    return lis_select_impl<bf16_t, float>(s, (const bf16_t*)h_physical, seg, sc, (char*)ws, (bf16_t*)out, idx, scores, l2p, p2l);
  }
  if (sc->wdtype == VSEL_BF16)
    return lis_select_impl<float, bf16_t>(s, (const float*)h_physical, seg, sc, (char*)ws, (float*)out, idx, scores, l2p, p2l);
  return lis_select_impl<float, float>(s, (const float*)h_physical, seg, sc, (char*)ws, (float*)out, idx, scores, l2p, p2l);
}

// Single-sweep form: the producer of the tokens already knows their column sums
// (vsel_gelu_colsum inside the merger + linearity of the merger's last Linear), so sweep 1 is skipped.
extern "C" int vsel_lis_select_presummed(void* stream, const void* h, vsel_dtype hdtype, const vsel_segments* seg,
                                         const vsel_scorer* sc, const float* col_sums, void* ws, size_t ws_bytes,
                                         const int64_t* logical_to_physical, const int64_t* physical_to_logical, void* out,
                                         int64_t* idx, float* scores) {
  LisPlan p;
  int st = lis_common_checks(h, seg, sc, hdtype, ws, ws_bytes, true, &p);
  if (st) return st;
  if (!out || !idx || !scores || !col_sums) return fail(VSEL_ERR_INVALID, "NULL pointer");
  if ((logical_to_physical == nullptr) != (physical_to_logical == nullptr))
    return fail(VSEL_ERR_INVALID, "give both row maps or neither");
  if (((uintptr_t)out | (uintptr_t)col_sums) & 15) return fail(VSEL_ERR_INVALID, "out / col_sums must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  VSEL_PROF_BEGIN(s);
  const int64_t* l2p = logical_to_physical;
  const int64_t* p2l = physical_to_logical;
  if (hdtype == VSEL_BF16) {
    if (sc->wdtype == VSEL_BF16)
      return lis_select_impl<bf16_t, bf16_t>(s, (const bf16_t*)h, seg, sc, (char*)ws, (bf16_t*)out, idx, scores, l2p, p2l, col_sums);
    return lis_select_impl<bf16_t, float>(s, (const bf16_t*)h, seg, sc, (char*)ws, (bf16_t*)out, idx, scores, l2p, p2l, col_sums);
  }
  if (sc->wdtype == VSEL_BF16)
    return lis_select_impl<float, bf16_t>(s, (const float*)h, seg, sc, (char*)ws, (float*)out, idx, scores, l2p, p2l, col_sums);
  return lis_select_impl<float, float>(s, (const float*)h, seg, sc, (char*)ws, (float*)out, idx, scores, l2p, p2l, col_sums);
}
