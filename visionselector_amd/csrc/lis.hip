// C-ABI entry points of the LIS inference path (kernels in lis_kernels.h).
#include "lis_kernels.h"

namespace vsel {
int check_segments(const vsel_segments* seg, bool need_k) { return check_segments_impl(seg, need_k); }
}  // namespace vsel

using namespace vsel;

extern "C" size_t vsel_lis_workspace_bytes(const vsel_segments* seg, int64_t d, int64_t hd) {
  if (!seg || seg->n_seg < 1 || d < 1 || hd < 1) return 0;
  return make_plan(seg->n_seg, seg->rows_per_seg, d, hd).total;
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
  hipStream_t s = (hipStream_t)stream;
  VSEL_PROF_BEGIN(s);
  if (hdtype == VSEL_BF16) {
    st = run_scores_w<bf16_t>(s, (const bf16_t*)h, seg, sc, (char*)ws, p, scores);
    if (st) return st;
    st = launch_select(s, scores, seg, idx, nullptr);
    if (st) return st;
    return launch_gather<bf16_t>(s, (const bf16_t*)h, (int)sc->d, seg, idx, (bf16_t*)out);
  }
  st = run_scores_w<float>(s, (const float*)h, seg, sc, (char*)ws, p, scores);
  if (st) return st;
  st = launch_select(s, scores, seg, idx, nullptr);
  if (st) return st;
  return launch_gather<float>(s, (const float*)h, (int)sc->d, seg, idx, (float*)out);
}
