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

__global__ __launch_bounds__(256, 1) void attn_fwd_gqa64_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                                const uint16_t* __restrict__ v, const int32_t* __restrict__ cu,
                                                                int hq, int hkv, float scale_log2e, int causal,
                                                                uint16_t* __restrict__ out, int q_tiles, int n_seq, int slot,
                                                                PagedKV pg, float* __restrict__ lse) {
  __shared__ __attribute__((aligned(1024))) char smem[kGqa64Lds + 16];
  int& s_item = *reinterpret_cast<int*>(smem + kGqa64Lds);
  const int rep = hq / hkv;
  const int wps = (rep + 1) >> 1;                   // waves per 32-query slice (two heads per wave)
  const int QW = 4 / wps;                           // slices per tile
  const int kBlockQ = 32 * QW;
  const int n_items = q_tiles * hkv * n_seq;
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
  const int hl = head_wave ? 2 * (wave % wps) : 0;  // first head of this wave inside the group
  const bool has_b = head_wave && hl + 1 < rep;     // (odd group sizes: the last wave of a slice has one head; block B replays it, nothing stored)
  const int lds_base = (int)lds_u32(smem);

#ifdef VSEL_GQA64_TRACE
  const unsigned long long t_kernel0 = __builtin_readcyclecounter();
  unsigned long long t_prev = t_kernel0;
#define GQA64_TRACE_END() do { if (tid == 0) atomicAdd(&g_gqa64_dbg[16 * (blockIdx.x & 63) + 12], (unsigned)(__builtin_readcyclecounter() - t_kernel0)); } while (0)
#else
#define GQA64_TRACE_END() do {} while (0)
#endif
  for (int round = 0;; ++round) {
    int item;
    if (slot == -2) {
      item = static_deal_item(round);
      if (item >= n_items) { GQA64_TRACE_END(); return; }
    } else if (slot < 0) {
      if (round > 0) { GQA64_TRACE_END(); return; }
      item = blockIdx.x;
      if (item >= n_items) return;
    } else {
      item = global_queue_next(&g_gqa64_work_counter[8 * (slot & 0xff)], n_items, &s_item, tid);
    }
    item = __builtin_amdgcn_readfirstlane(item);
    if (item < 0) { GQA64_TRACE_END(); return; }
    const int n_pairs = hkv * n_seq;
    const int level = item / n_pairs, pair = item - level * n_pairs;
    const int seq = pair / hkv, kvh = pair - seq * hkv;
    const int qs = cu[seq];
    const int qlen = cu[seq + 1] - qs;
    const int q0 = (q_tiles - 1 - level) * kBlockQ;
    if (q0 >= qlen) {
      if (slot >= 0 && (slot & 0x200) && kvh == 0)
        queue_skip_empty_run(&g_gqa64_work_counter[8 * (slot & 0xff)], tid, cu, n_seq, hkv, level, seq,
                             [&](int lv, int ql) { return (q_tiles - 1 - lv) * kBlockQ < ql; });
      continue;
    }
    const int len = qlen;
    const int wave_qmin = q0 + 32 * slice;
    const int wave_qmax = min(wave_qmin + 31, qlen - 1);
    const int my_q = min(wave_qmin + j, qlen - 1);
    const bool row_ok = head_wave && wave_qmin + j < qlen;
    const int kv_end = causal ? min(len, q0 + kBlockQ) : len;
    const int n_tiles = (kv_end + kTileK - 1) / kTileK;
    int n_w = (head_wave && wave_qmin < qlen) ? n_tiles : 0;
    if (causal && n_w > 0) n_w = min(n_tiles, wave_qmax / kTileK + 1);
    n_w = __builtin_amdgcn_readfirstlane(n_w);
    int mfirst = len / kTileK;
    if (causal) mfirst = min(mfirst, wave_qmin / kTileK + ((wave_qmin % kTileK) != kTileK - 1 ? 0 : 1));
    mfirst = __builtin_amdgcn_readfirstlane(mfirst);
    const int kmax = causal ? min(len - 1, my_q) : len - 1;
    const int head_a = kvh * rep + hl;
    const uint16_t* const qbase = uniform_ptr64(q + (int64_t)qs * hq * kHeadDim64 + (int64_t)wave_qmin * q_rs + head_a * q_hs);
    const uint16_t* const kbase = uniform_ptr64(k + (int64_t)qs * hkv * kHeadDim64 + kvh * kv_hs);
    const uint16_t* const vbase = uniform_ptr64(v + (int64_t)qs * hkv * kHeadDim64 + kvh * v_hs);
    const uint16_t* const obase = uniform_ptr64(out + ((int64_t)(qs + wave_qmin) * hq + head_a) * kHeadDim64);
    const int qrs2 = (int)(q_rs * 2), krs2 = (int)(kv_rs * 2), vrs2 = (int)(v_rs * 2);
    const int qhs2 = has_b ? (int)(q_hs * 2) : 0;
    const int ostride = hq * kHeadDim64 * 2;
    const int nvalid = __builtin_amdgcn_readfirstlane(head_wave ? qlen - wave_qmin : 0);
    const int nvalidb = has_b ? nvalid : 0;
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
                   [qhs2] "s"(qhs2), [ohs2] "s"(2 * kHeadDim64), [nvalidb] "s"(nvalidb) VSEL_GQA64_DBG_OPERAND
                 : VSEL_GQA64_ASM_CLOBBERS);
    if (lse) {
      const float lt0 = l0 + __shfl_xor(l0, 32, 64), lt1 = l1 + __shfl_xor(l1, 32, 64);
      if (hh == 0 && row_ok) {
        const int64_t r = (int64_t)(qs + my_q) * hq + head_a;
        lse[r] = lt0 > 0.f ? (m0 + log2f(lt0)) * 0.6931471805599453f : -INFINITY;
        if (has_b) lse[r + 1] = lt1 > 0.f ? (m1 + log2f(lt1)) * 0.6931471805599453f : -INFINITY;
      }
    }
    __syncthreads();                   // the next item's first loads overwrite ring slots / staging rows other waves may still read
#ifdef VSEL_GQA64_TRACE
    t_prev = __builtin_readcyclecounter();
#endif
  }
}

namespace attn {

int attn_fwd_gqa64_launch(hipStream_t st, const void* q, const void* k, const void* v, const int32_t* cu_q, int64_t n_seq,
                          int64_t max_seqlen_q, int64_t hq, int64_t hkv, float scale, int causal, void* out, const PagedKV& pg, float* lse) {
  const int rep = (int)(hq / hkv);
  const int block_q = 32 * (4 / ((rep + 1) / 2));
  const int q_tiles = (int)cdiv(max_seqlen_q, block_q);
  const int64_t n_items = (int64_t)q_tiles * hkv * n_seq;
  if (n_items >= (1ll << 31)) return fail(VSEL_ERR_UNSUPPORTED, "too many attention work items");
  const int64_t slots = 256;
  int slot = -1, taken = -1;
  if (attn_static_deal(n_items, slots, true, 36)) {
    slot = -2;
  } else if (n_items > slots) {
    if (int rc = queue_slot_acquire(kSlotGqa, st, &taken)) return rc;
    slot = taken;
    int* counters = nullptr;
    VSEL_HIP_CHECK(hipGetSymbolAddress((void**)&counters, HIP_SYMBOL(g_gqa64_work_counter)));
    VSEL_HIP_CHECK(hipMemsetAsync(counters + 8 * slot, 0, 8 * sizeof(int), st));
    if (n_seq > 1 && knob(VSEL_KNOB_ATTN_SKIP_EMPTY) != 0) slot |= 0x200;
  }
  const dim3 grid((unsigned)std::min<int64_t>(n_items, slots));
  hipLaunchKernelGGL(attn_fwd_gqa64_kernel, grid, dim3(256), 0, st, (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v, cu_q,
                     (int)hq, (int)hkv, scale * 1.4426950408889634f, causal, (uint16_t*)out, q_tiles, (int)n_seq, slot, pg, lse);
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
