// C-ABI entry points of the LIS inference path (kernels in lis_kernels.h).
#include "lis_kernels.h"
#include "lis_small.h"

#include <stdlib.h>

namespace vsel {
int check_segments(const vsel_segments* seg, bool need_k) { return check_segments_impl(seg, need_k); }
}  // namespace vsel

using namespace vsel;

// ---- two-half software pipeline ----------------------------------------------------------------------------------
// The path is bound by the fabric between the XCDs and memory (HBM and Infinity-Cache traffic are NOT additive on this part,
// tools/membw3.hip), so the only time to win at large batch is the ~70 us of small latency-bound kernels between the sweeps
// (projections, finish kernels, radix select).  With >= kPipelineMinSegments segments the batch is cut in two halves A, B that
// run on the caller's stream and on a library-owned auxiliary stream, so the small kernels of one half execute underneath the
// sweeps of the other:
//
//   caller's stream : S1(A) S1(B) --fork--> proj(B) S2(B) select(B) G(B) <--join--
//   auxiliary stream:                       proj(A) S2(A) select(A) G(A)
//
// Exactly two cross-stream hand-offs per call: a hand-off costs ~8 us on this runtime, and a schedule that kept the sweeps on
// one stream and only the small kernels on the other (8 hand-offs) measured 5 % SLOWER than no pipeline at all.  Every kernel
// is deterministic and each half is self-contained with a batch-invariant plan, so the results are bit-identical to the
// single-stream order.  Under vsel_profile_start() the same half-sized launches run back to back on the caller's stream, so
// per-kernel HIP-event durations refer to the launches the product path really makes.
constexpr int64_t kPipelineMinSegments = 32;
constexpr int kEventSets = 8, kEventsPerSet = 8;

struct AuxStream {
  hipStream_t stream = nullptr;
  hipEvent_t ev[kEventSets][kEventsPerSet];
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
    for (int i = 0; i < kEventSets; ++i)
      for (int j = 0; j < kEventsPerSet; ++j)
        if (hipEventCreateWithFlags(&a.ev[i][j], hipEventDisableTiming) != hipSuccess) return nullptr;
    a.ok = true;
  }
  return &a;
}

// OFF by default (VSEL_PIPELINE=1 or the knob VSEL_KNOB_LIS_PIPELINE, include/vsel_debug.h, turns it on): measured on MI355X at B = 32 / 64 / 128 images it is
// 264.7 / 454.3 / 884.6 us per call against 240.9 / 449.2 / 878.6 us for the single-piece, single-stream order -- the halves pay
// the latency-bound small kernels twice and the two hand-offs, which eats what the overlap wins.
static inline int pipeline_enabled() { return knob(VSEL_KNOB_LIS_PIPELINE); }

// small-batch form (lis_small.h): used up to small_path_max_seg() segments per call.  Measured on MI355X (Qwen2.5-VL-7B geometry,
// us per call, small form vs batched form): 1 image 36.2 vs 51.7, 2 images 43.8 vs 55.8, 3 images 52.2 vs 60.8, 4 images 60.4 vs
// 65.3, 8 images 102.3 vs 86.2 -- the redundant prologues grow with the segment count, so the default limit is 4.
// VSEL_SMALL_PATH=<n> / knob VSEL_KNOB_LIS_SMALL_PATH set the limit (0 = never; at most kSmallMaxSeg).
static_assert(kSmallMaxSeg == 8, "common.hip clamps VSEL_KNOB_LIS_SMALL_PATH to 8");
static inline int small_path_max_seg() { return knob(VSEL_KNOB_LIS_SMALL_PATH); }
#ifdef VSEL_TRACE
// copies the stamps of the last small-batch call: out[kTraceKernels][kTraceBlocks][kTraceSlots] (tools/trace_small.py)
extern "C" int vsel_debug_read_trace(unsigned long long* out, int clear) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(g_trace)) != hipSuccess) return VSEL_ERR_HIP;
  if (clear) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_trace)) != hipSuccess || hipMemset(p, 0, sizeof(g_trace)) != hipSuccess)
      return VSEL_ERR_HIP;
  }
  return VSEL_OK;
}
#endif

// select fused into the gather (select_gather_small_kernel) for mid-size batches of the nine-launch form.  Measured (7B geometry,
// us per call, fused vs two launches): 8 images 82.0 vs 83.3, 16 135.3 vs 137.0, 32 233.4 vs 238.8, 48 350.9 vs 347.1, 128 900.8 vs
// 877.0 -> up to 32 segments; re-measured at the end of round 3: 16 136.9 vs 138.5, 32 227.7 vs 232.9, 48 347.4 vs 353.7, 64 428.4 vs
// 428.7 -> up to 48.  VSEL_FUSED_SELECT=<n> / knob VSEL_KNOB_LIS_FUSED_SELECT.
static inline int fused_select_max_seg() { return knob(VSEL_KNOB_LIS_FUSED_SELECT); }

static bool use_small_path(const vsel_segments* seg, const vsel_scorer* sc, const LisPlan& p) {
  return seg->n_seg <= small_path_max_seg() && small_path_ok(seg, sc, p);
}

template <typename T, typename TW>
static int select_whole(hipStream_t st, const T* h, const vsel_segments* seg, const vsel_scorer* sc, char* ws, const LisPlan& p,
                        T* out, int64_t* idx, float* scores, const int64_t* l2p = nullptr, const int64_t* p2l = nullptr,
                        const float* col_sums = nullptr) {
  int rc = VSEL_OK;
  if (!col_sums) rc = run_colsum<T>(st, h, seg, (int)sc->d, ws, p);
  if (rc) return rc;
  if (use_small_path(seg, sc, p)) {
    // a handful of segments: five launches instead of nine, bit-identical results (lis_small.h)
    if ((rc = run_proj_small(st, seg, sc, ws, p, col_sums))) return rc;
    if ((rc = run_score_small<T>(st, h, seg, sc, ws, p, scores, p2l))) return rc;
    return launch_select_gather_small<T>(st, h, (int)sc->d, seg, scores, idx, out, l2p);
  }
  rc = run_proj<TW>(st, seg, sc, ws, p, col_sums);
  if (rc) return rc;
  rc = run_score<T>(st, h, seg, sc, ws, p, scores, p2l);
  if (rc) return rc;
  // up to fused_select_max_seg() segments: the radix select runs inside every gather workgroup (one launch less; same indices)
  // (every gather workgroup repeats its segment's select: not when there are thousands of them -- 32 images at 50 % retain:
  // 308.0 vs 305.7 us)
  if (seg->n_seg <= fused_select_max_seg() && seg->n_seg * cdiv(seg->k, 16) <= 1536)
    return launch_select_gather_small<T>(st, h, (int)sc->d, seg, scores, idx, out, l2p);
  rc = launch_select(st, scores, seg, idx, nullptr);
  if (rc) return rc;
  return launch_gather<T>(st, h, (int)sc->d, seg, idx, out, l2p);
}

template <typename T, typename TW>
static int lis_select_impl(hipStream_t st, const T* h, const vsel_segments* seg, const vsel_scorer* sc, char* ws, T* out,
                           int64_t* idx, float* scores, const int64_t* l2p = nullptr, const int64_t* p2l = nullptr,
                           const float* col_sums = nullptr) {
  const int64_t S = seg->n_seg, d = sc->d;
  const bool halves = pipeline_enabled() && S >= kPipelineMinSegments && !l2p && !col_sums;
  if (!halves) {
    const LisPlan p = make_plan(S, seg->rows_per_seg, d, sc->hd);
    return select_whole<T, TW>(st, h, seg, sc, ws, p, out, idx, scores, l2p, p2l, col_sums);
  }
  // halves A = [0, s0), B = [s0, S).  Uniform segments address rows relative to the half's base pointer; ragged
  // segments keep absolute offsets (seg_rows / seg_out are advanced instead).
  const int64_t s0 = S / 2;
  vsel_segments sg[2] = {*seg, *seg};
  sg[0].n_seg = s0;
  sg[1].n_seg = S - s0;
  const T* hh[2] = {h, h};
  T* oo[2] = {out, out};
  int64_t* ii[2] = {idx, idx};
  float* ss[2] = {scores, scores};
  if (seg->seg_rows) {
    sg[1].seg_rows = seg->seg_rows + s0;
    sg[1].seg_out = seg->seg_out + s0;
    sg[0].total_rows = sg[1].total_rows = seg->total_rows;   // informational only in the ragged form
  } else {
    sg[0].total_rows = s0 * seg->rows_per_seg;
    sg[0].total_out = s0 * seg->k;
    sg[1].total_rows = (S - s0) * seg->rows_per_seg;
    sg[1].total_out = (S - s0) * seg->k;
    hh[1] = h + sg[0].total_rows * d;
    oo[1] = out + sg[0].total_out * d;
    ii[1] = idx + sg[0].total_out;
    ss[1] = scores + sg[0].total_rows;
  }
  const PolicyScope whole_call_policy(seg->total_rows);   // cache policy of the sweeps = that of the whole call
  const LisPlan pl[2] = {make_plan(sg[0].n_seg, seg->rows_per_seg, d, sc->hd), make_plan(sg[1].n_seg, seg->rows_per_seg, d, sc->hd)};
  char* wss[2] = {ws, ws + pl[0].total};
  AuxStream* aux = prof_enabled() ? nullptr : aux_for_current_device();
  hipStream_t sx = aux ? aux->stream : st;          // where the small kernels go
  hipEvent_t* ev = aux ? aux->ev[aux->next++ % kEventSets] : nullptr;
  auto hand = [&](int e, hipStream_t from, hipStream_t to) -> int {   // "to" continues after what "from" has queued so far
    if (!aux) return VSEL_OK;
    VSEL_HIP_CHECK(hipEventRecord(ev[e], from));
    VSEL_HIP_CHECK(hipStreamWaitEvent(to, ev[e], 0));
    return VSEL_OK;
  };
  int rc;
  for (int x = 0; x < 2; ++x)                       // S1(A) S1(B)
    if ((rc = run_colsum<T>(st, hh[x], &sg[x], (int)d, wss[x], pl[x]))) return rc;
  if ((rc = hand(0, st, sx))) return rc;            // fork: half A continues on the auxiliary stream
  for (int x = 1; x >= 0; --x) {                    // the rest of B on the caller's stream, the rest of A on the auxiliary one
    hipStream_t s = x == 1 ? st : sx;
    if ((rc = run_proj<TW>(s, &sg[x], sc, wss[x], pl[x]))) return rc;
    if ((rc = run_score<T>(s, hh[x], &sg[x], sc, wss[x], pl[x], ss[x]))) return rc;
    if ((rc = launch_select(s, ss[x], &sg[x], ii[x], nullptr))) return rc;
    if ((rc = launch_gather<T>(s, hh[x], (int)d, &sg[x], ii[x], oo[x]))) return rc;
  }
  if ((rc = hand(1, sx, st))) return rc;            // join
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
  if (use_small_path(seg, sc, p)) {
    int rc;
    if (hdtype == VSEL_BF16) {
      if ((rc = run_colsum<bf16_t>(s, (const bf16_t*)h, seg, (int)sc->d, (char*)ws, p))) return rc;
      if ((rc = run_proj_small(s, seg, sc, (char*)ws, p, nullptr))) return rc;
      return run_score_small<bf16_t>(s, (const bf16_t*)h, seg, sc, (char*)ws, p, scores, nullptr);
    }
    if ((rc = run_colsum<float>(s, (const float*)h, seg, (int)sc->d, (char*)ws, p))) return rc;
    if ((rc = run_proj_small(s, seg, sc, (char*)ws, p, nullptr))) return rc;
    return run_score_small<float>(s, (const float*)h, seg, sc, (char*)ws, p, scores, nullptr);
  }
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
