// Group-shared short-sequence attention forward on the hand-scheduled 64-rows-per-wave loop (gfx950, bf16, head_dim 128).
//
// attn_fwd64.hip's generated per-item body -- one 512-register wave per SIMD, S(t + 1) beside the exponentials of tile t, P V beside the
// maxima, K two tiles ahead, every K / V fragment read from LDS once for TWO 32-row blocks -- with the blocks re-assigned: instead of 64
// consecutive queries of one head a wave owns the SAME 32 queries of TWO q heads of a GQA group (generator option heads = 1,
// tools/gen_attn_fwd64.py -> attn_fwd_gqa64_body.inc).  A workgroup (4 waves) therefore serves up to 8 q heads of ONE kv head on a
// 32-query tile: K / V tiles are loaded once per group (attn_fwd_gqa.hip's sharing) and the tile loop costs ~8 instructions per MFMA
// instead of the ~18 of the compiler-scheduled 32-rows-per-wave forms, which are instruction-issue bound (profiles/r05_attn_pmc.txt).
// Same arithmetic per query row as every other form: outputs and log-sum-exps are bit-identical (tests force all forms).
// Reference call sites: qwen-evaluation/qwen25vl/modeling_qwen2_5_vl.py:827-918, qwen-vl-finetune/qwenvl/train/trainer.py:101-113.
#include "attn_common.h"
#include <atomic>
#ifndef VSEL_GQA64_BODY
#define VSEL_GQA64_BODY "attn_fwd_gqa64_body.inc"      // (tools/trace_gqa64.py compiles a trace = 1 body)
#endif
#include VSEL_GQA64_BODY

#include <algorithm>

namespace vsel {

using namespace attn;

namespace {
__device__ __forceinline__ const uint16_t* uniform_ptr64(const uint16_t* p) {
  const uint64_t u = (uint64_t)(uintptr_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return (const uint16_t*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
constexpr int kGqa64Lds = VSEL_GQA64_LDS_BYTES;
constexpr int kHeadDim64 = 128;
}  // namespace

__device__ int g_gqa64_work_counter[64 * 8];
#ifdef VSEL_GQA64_TRACE
// trace builds (generator option trace = 1): shader cycles per wave summed over waves and items, one 64-byte line per workgroup slot:
// [0] step-top wait + barrier, [1] phase X, [2] phase Y, [3] exponent tail, [4] non-steady steps, [5] prologue rest, [6] epilogue,
// [7] steady steps, [8] prologue issue, [9] prologue wait + barrier, [10] S(0), [12] whole kernel (thread 0), [13] items, [14] queue + setup
__device__ unsigned g_gqa64_dbg[64 * 16];
#define VSEL_GQA64_DBG_OPERAND , [dbg] "s"(&g_gqa64_dbg[16 * (blockIdx.x & 63)])
#else
#define VSEL_GQA64_DBG_OPERAND
#endif

struct Gqa64Item {
  int seq, kvh, q0, qs, qlen, n_tiles;             // n_tiles == 0: not an item (empty level of a shorter sequence, or past the end of the list)
};

// Items are pipelined into each other (generator option xitem): while an item's last tile is computed the NEXT item's Q rows, K(0), K(1) and
// V(0) are loaded (in the gaps of the P V MFMAs), its Q fragments are read before the epilogue, and the work queue is drawn one item further
// still -- a workgroup always knows its next item when it starts one.
__global__ __launch_bounds__(256, 1) void attn_fwd_gqa64_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                                const uint16_t* __restrict__ v, const int32_t* __restrict__ cu,
                                                                int hq, int hkv, float scale_log2e, int causal,
                                                                uint16_t* __restrict__ out, int q_tiles, int n_seq, int* __restrict__ counter,
                                                                int static_deal, PagedKV pg, float* __restrict__ lse) {
  __shared__ __attribute__((aligned(1024))) char smem[kGqa64Lds + 16];
  int& s_item = *reinterpret_cast<int*>(smem + kGqa64Lds);
  const int rep = hq / hkv;
  const int wps = (rep + 1) >> 1;                   // waves per 32-query slice (two heads per wave)
  const int QW = 4 / wps;                           // slices per tile
  const int kBlockQ = 32 * QW;
  const int n_pairs = hkv * n_seq;
  const int n_items = q_tiles * n_pairs;
  const int64_t q_rs = pg.q_row_stride ? pg.q_row_stride : (int64_t)hq * kHeadDim64;
  const int64_t q_hs = pg.q_row_stride ? pg.q_head_stride : kHeadDim64;
  const int64_t kv_rs = pg.kv_row_stride ? pg.kv_row_stride : (int64_t)hkv * kHeadDim64;
  const int64_t kv_hs = pg.kv_row_stride ? pg.kv_head_stride : kHeadDim64;
  const int64_t v_rs = pg.v_row_stride ? pg.v_row_stride : kv_rs;
  const int64_t v_hs = pg.v_row_stride ? pg.v_head_stride : kv_hs;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;
  const bool head_wave = wave < QW * wps;           // (5 / 6 heads: three waves per slice, the fourth only helps to load)
  const int slice = head_wave ? wave / wps : 0;
  const int hl = head_wave ? 2 * (wave % wps) : 0;  // first head of this wave inside the group (a helper wave replays head 0's rows, stores nothing)
  const bool has_b = head_wave && hl + 1 < rep;     // (odd group sizes: the last wave of a slice has one head; block B replays it, nothing stored)
  const int lds_base = (int)lds_u32(smem);
  const int qrs2 = (int)(q_rs * 2), krs2 = (int)(kv_rs * 2), vrs2 = (int)(v_rs * 2);
  const int qhs2 = has_b ? (int)(q_hs * 2) : 0;
  const int ostride = hq * kHeadDim64 * 2;
#ifdef VSEL_GQA64_TRACE
  const unsigned long long t_kernel0 = __builtin_readcyclecounter();
  unsigned long long t_prev = t_kernel0;
#define GQA64_TRACE_END() do { if (tid == 0) atomicAdd(&g_gqa64_dbg[16 * (blockIdx.x & 63) + 12], (unsigned)(__builtin_readcyclecounter() - t_kernel0)); } while (0)
#else
#define GQA64_TRACE_END() do {} while (0)
#endif

  auto decode = [&](int item) -> Gqa64Item {
    Gqa64Item it{0, 0, 0, 0, 0, 0};
    if (item >= n_items) return it;
    const int level = item / n_pairs, pair = item - level * n_pairs;
    it.seq = pair / hkv;
    it.kvh = pair - it.seq * hkv;
    it.qs = cu[it.seq];
    it.qlen = cu[it.seq + 1] - it.qs;
    it.q0 = (q_tiles - 1 - level) * kBlockQ;
    if (it.q0 >= it.qlen) return it;
    const int kv_end = causal ? min(it.qlen, it.q0 + kBlockQ) : it.qlen;
    it.n_tiles = (kv_end + kTileK - 1) / kTileK;
    return it;
  };
  // a candidate that is not an item: the counter jumps over the run of empty items it starts (attn_common.h), draw again
  auto validate = [&](int cand) -> Gqa64Item {
    for (;;) {
      Gqa64Item it = decode(cand);
      if (it.n_tiles > 0 || cand >= n_items || !counter) return it;
      const int level = cand / n_pairs, pair = cand - level * n_pairs;
      if (pair % hkv == 0)
        queue_skip_empty_run(counter, tid, cu, n_seq, hkv, level, pair / hkv, [&](int lv, int ql) { return (q_tiles - 1 - lv) * kBlockQ < ql; });
      if (tid == 0) s_item = atomicAdd(counter, 1);
      __syncthreads();
      cand = __builtin_amdgcn_readfirstlane(s_item);
      __syncthreads();
    }
  };
  auto q_ptr = [&](const Gqa64Item& it) {
    return uniform_ptr64(q + (int64_t)it.qs * hq * kHeadDim64 + (int64_t)(it.q0 + 32 * slice) * q_rs + (it.kvh * rep + hl) * q_hs);
  };
  auto k_ptr = [&](const Gqa64Item& it) { return uniform_ptr64(k + (int64_t)it.qs * hkv * kHeadDim64 + it.kvh * kv_hs); };
  auto v_ptr = [&](const Gqa64Item& it) { return uniform_ptr64(v + (int64_t)it.qs * hkv * kHeadDim64 + it.kvh * v_hs); };
  // what the previous body may load for an item: whole tiles and 32 whole rows in every slice (the body's in-gap loads carry no clamps)
  auto prefetchable = [&](const Gqa64Item& it) {
    return it.n_tiles > 0 && it.q0 + kBlockQ <= it.qlen && it.qlen >= (it.n_tiles > 1 ? 2 * kTileK : kTileK);
  };

  // ---- the first item is the workgroup's own index, the second is drawn (and waited for) once ---------------------------------------------
  // (static_deal: few rounds of items -- workgroup b takes items b, 2 G - 1 - b, 2 G + b, ... of the heaviest-first list, attn_common.h: no
  // atomic, no publish; empty items of ragged batches are simply skipped)
  int deal_round = 0;
  auto deal_next = [&]() -> Gqa64Item {
    for (;;) {
      const int it_idx = static_deal_item(++deal_round);
      const Gqa64Item it = decode(it_idx);
      if (it.n_tiles > 0 || it_idx >= n_items) return it;
    }
  };
  Gqa64Item cur = decode((int)blockIdx.x);
  Gqa64Item next{0, 0, 0, 0, 0, 0};
  if (static_deal) {
    if (cur.n_tiles == 0) cur = deal_next();
    if (cur.n_tiles == 0) return;
    next = deal_next();
  } else if (counter) {
    // the second item is dealt too (2 G - 1 - b, the mirror of the first round; the counter starts at 2 G): with every draw an item ahead the
    // queue would hand the second round out in arrival order, and the workgroup on a level's heaviest item would get the next level's heaviest.
    // An EMPTY first item (ragged batches) hands its slot to the dealt one; the queue then supplies the second.
    const bool first_empty = cur.n_tiles == 0;
    if (first_empty) cur = validate(static_deal_item(1));
    if (cur.n_tiles == 0) return;
    if (first_empty) {
      if (tid == 0) s_item = atomicAdd(counter, 1);
      __syncthreads();
      const int cand = __builtin_amdgcn_readfirstlane(s_item);
      __syncthreads();
      next = validate(cand);
    } else {
      next = validate(static_deal_item(1));
    }
  } else if (cur.n_tiles == 0) {
    return;
  }
  int pref = 0;                                     // what the previous body already loaded of `cur`: bit 0 Q fragments + K(0) + K(1), bit 1 V(0)
  for (;;) {
    int pend = 0;                                   // thread 0: the draw for the item after next, in flight during the body
    if (counter && next.n_tiles > 0 && tid == 0) pend = atomicAdd(counter, 1);
    const int qlen = cur.qlen, len = cur.qlen;
    const int wave_qmin = cur.q0 + 32 * slice;
    const int wave_qmax = min(wave_qmin + 31, qlen - 1);
    const int my_q = min(wave_qmin + j, qlen - 1);
    const bool row_ok = head_wave && wave_qmin + j < qlen;
    const int n_tiles = cur.n_tiles;
    int n_w = (head_wave && wave_qmin < qlen) ? n_tiles : 0;
    if (causal && n_w > 0) n_w = min(n_tiles, wave_qmax / kTileK + 1);
    n_w = __builtin_amdgcn_readfirstlane(n_w);
    int mfirst = len / kTileK;
    if (causal) mfirst = min(mfirst, wave_qmin / kTileK + ((wave_qmin % kTileK) != kTileK - 1 ? 0 : 1));
    mfirst = __builtin_amdgcn_readfirstlane(mfirst);
    const int kmax = causal ? min(len - 1, my_q) : len - 1;
    const int head_a = cur.kvh * rep + hl;
    const uint16_t* const qbase = q_ptr(cur);
    const uint16_t* const kbase = k_ptr(cur);
    const uint16_t* const vbase = v_ptr(cur);
    const uint16_t* const obase = uniform_ptr64(out + ((int64_t)(cur.qs + wave_qmin) * hq + head_a) * kHeadDim64);
    const int nvalid = __builtin_amdgcn_readfirstlane(head_wave ? qlen - wave_qmin : 0);
    // the next item's loads ride in this item's last tile: K slots are both free there, V slot 0 only when that tile sits in slot 1
    const bool nx_ok = prefetchable(next);
    const int nx_flags = nx_ok ? (1 | (((n_tiles - 1) & 1) ? 2 : 0)) : 0;
    const uint16_t* const nxq = nx_ok ? q_ptr(next) : qbase;
    const uint16_t* const nxk0 = nx_ok ? k_ptr(next) : kbase;
    const uint16_t* const nxk1 = (nx_ok && next.n_tiles > 1) ? nxk0 + (int64_t)kTileK * kv_rs : nxk0;
    const uint16_t* const nxv = nx_ok ? v_ptr(next) : vbase;
    const int flags = __builtin_amdgcn_readfirstlane(pref | (has_b ? 4 : 0) | (nx_flags << 3));
    const int len_u = __builtin_amdgcn_readfirstlane(len), ntiles_u = __builtin_amdgcn_readfirstlane(n_tiles);
    float m0, m1, l0, l1;
#ifdef VSEL_GQA64_TRACE
    if (lane == 0) {                                  // everything between two bodies: draw, decode, pointer set-up, log-sum-exp, item barrier
      const unsigned long long now = __builtin_readcyclecounter();
      atomicAdd(&g_gqa64_dbg[16 * (blockIdx.x & 63) + 14], (unsigned)(now - t_prev));
      if (wave == 0) atomicAdd(&g_gqa64_dbg[16 * (blockIdx.x & 63) + 13], 1u);
    }
#endif
    asm volatile(VSEL_GQA64_ASM_TEXT
                 : [m0] "=&v"(m0), [m1] "=&v"(m1), [l0] "=&v"(l0), [l1] "=&v"(l1)
                 : [qbase] "s"(qbase), [qrs2] "s"(qrs2), [obase] "s"(obase), [ostride] "s"(ostride), [nvalid] "s"(nvalid), [kbase] "s"(kbase),
                   [vbase] "s"(vbase), [krs2] "s"(krs2), [vrs2] "s"(vrs2), [ntiles] "s"(ntiles_u), [nw] "s"(n_w), [mfirst] "s"(mfirst),
                   [len] "s"(len_u), [c] "s"(scale_log2e), [wave] "s"(wave), [ldsbase] "s"(lds_base), [kmaxa] "v"(kmax), [kmaxb] "v"(kmax),
                   [qhs2] "s"(qhs2), [ohs2] "s"(2 * kHeadDim64), [flags] "s"(flags), [nxq] "s"(nxq), [nxk0] "s"(nxk0), [nxk1] "s"(nxk1),
                   [nxv] "s"(nxv) VSEL_GQA64_DBG_OPERAND
                 : VSEL_GQA64_ASM_CLOBBERS);
    if (lse) {
      const float lt0 = l0 + __shfl_xor(l0, 32, 64), lt1 = l1 + __shfl_xor(l1, 32, 64);
      if (hh == 0 && row_ok) {
        const int64_t r = (int64_t)(cur.qs + my_q) * hq + head_a;
        lse[r] = lt0 > 0.f ? (m0 + log2f(lt0)) * 0.6931471805599453f : -INFINITY;
        if (has_b) lse[r + 1] = lt1 > 0.f ? (m1 + log2f(lt1)) * 0.6931471805599453f : -INFINITY;
      }
    }
    if (next.n_tiles == 0) { GQA64_TRACE_END(); return; }
    cur = next;
    pref = nx_flags;
    if (static_deal) {
      // the next item's own loads (what was not prefetched) overwrite ring slots other waves may still read -- nothing is left to load when
      // the last body prefetched all of it (Q, K(0), K(1), V(0): the next body's first step barrier is then the first point of contact)
      if (pref != 3) __syncthreads();
      next = deal_next();
    } else {
      if (tid == 0) s_item = pend;
      __syncthreads();                 // (the same barrier)
      const int cand = __builtin_amdgcn_readfirstlane(s_item);
      __syncthreads();                 // (s_item is rewritten by the next item's publish / the empty-candidate path)
      next = validate(cand);
    }
#ifdef VSEL_GQA64_TRACE
    t_prev = __builtin_readcyclecounter();
#endif
  }
}

namespace attn {

int attn_fwd_gqa64_launch(hipStream_t st, const void* q, const void* k, const void* v, const int32_t* cu_q, int64_t n_seq,
                          int64_t max_seqlen_q, int64_t hq, int64_t hkv, float scale, int causal, void* out, const PagedKV& pg, float* lse,
                          bool deal) {
  const int rep = (int)(hq / hkv);
  const int block_q = 32 * (4 / ((rep + 1) / 2));
  const int q_tiles = (int)cdiv(max_seqlen_q, block_q);
  const int64_t n_items = (int64_t)q_tiles * hkv * n_seq;
  if (n_items >= (1ll << 31)) return fail(VSEL_ERR_UNSUPPORTED, "too many attention work items");
  const int grid = (int)std::min<int64_t>(n_items, 256);
  int taken = -1;
  int* counter = nullptr;                          // one item per workgroup: no queue
  const int static_deal = (deal || attn_static_deal(n_items, 256, true, 36)) ? 1 : 0;     // (deal: uniform batches, any number of rounds)
  if (n_items > grid && !static_deal) {
    if (int rc = queue_slot_acquire(kSlotGqa, st, &taken)) return rc;
    int* counters = nullptr;
    VSEL_HIP_CHECK(hipGetSymbolAddress((void**)&counters, HIP_SYMBOL(g_gqa64_work_counter)));
    counter = counters + 8 * taken;
    VSEL_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)counter, 2 * grid, 1, st));     // workgroup b starts with items b and 2 G - 1 - b
  }
  VSEL_LAUNCH(attn_fwd_gqa64_kernel, dim3(grid), dim3(256), 0, st, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v, cu_q,
                     (int)hq, (int)hkv, scale * 1.4426950408889634f, causal, (uint16_t*)out, q_tiles, (int)n_seq, counter, static_deal, pg, lse);
  queue_slot_launched(kSlotGqa, taken, st);
  VSEL_AFTER_LAUNCH(st, "attn_fwd_gqa64_kernel");
  return VSEL_OK;
}

}  // namespace attn
}  // namespace vsel

#ifdef VSEL_GQA64_TRACE
extern "C" int vsel_debug_read_gqa64_trace(unsigned* out, int clear) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(vsel::g_gqa64_dbg), sizeof(vsel::g_gqa64_dbg)) != hipSuccess) return VSEL_ERR_HIP;
  if (clear) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(vsel::g_gqa64_dbg)) != hipSuccess || hipMemset(p, 0, sizeof(vsel::g_gqa64_dbg)) != hipSuccess) return VSEL_ERR_HIP;
  }
  return VSEL_OK;
}
#endif
