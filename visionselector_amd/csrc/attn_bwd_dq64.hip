// dQ pass of the var-len causal / full GQA attention backward, head_dim 128, bf16 -- 64 query rows per wave (gfx950).
//
// Same arithmetic, per query row and operation for operation, as attn_bwd_dq_kernel (attn_bwd.hip; reference: the autograd of the
// eager formula, qwen-evaluation/qwen25vl/modeling_qwen2_5_vl.py:777-797, reached in training through
// qwen-vl-finetune/qwenvl/train/trainer.py:101-113): dQ, and D = rowsum(dO o O) and lse * log2(e) left in the workspace for the dK / dV
// kernel, are bit-identical to that kernel's.  The structure is the 64-rows-per-wave forward's (attn_fwd64.hip): four waves, one per SIMD,
// each owning 64 queries and its SIMD's whole register file -- dQ^T (128), Q^T (64) and dO^T (64) fragments in accumulator registers,
// S / dP / dS and the K, V, K^T fragments in arch VGPRs -- every K / V row fragment... feeding its MFMA from a two-slot direct-to-LDS ring,
// one barrier per 64-key tile, and the MFMA stream software-pipelined inside the wave in units of (32-key block, 32-row block): the
// exponentials and dS of a unit run in the gaps of the other row block's MFMAs (tools/gen_attn_bwd_dq64.py has the stream).  Q, dO and O
// rows arrive by whole-row direct-to-LDS loads, dQ leaves as whole rows through the ring.
// The per-item body is one GENERATED inline-asm statement (hipcc cannot hold the register plan: attn_fwd64.hip).
#include "attn_common.h"
#include <atomic>
#ifndef VSEL_DQ64_BODY
#define VSEL_DQ64_BODY "attn_bwd_dq64_body.inc"
#endif
#include VSEL_DQ64_BODY

#include <algorithm>

namespace vsel {

using namespace attn;

namespace {
__device__ __forceinline__ const void* uniform_ptr(const void* p) {
  const uint64_t u = (uint64_t)(uintptr_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return (const void*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
constexpr int kLds = 2 * 65536;             // the K[2] + V[2] ring (also staging of dO rows and of the dQ tile) + 4 x 16 KiB of row staging
constexpr int kBlockQ = 256;
constexpr int kD = 128;
}  // namespace

__device__ int g_dq64_work_counter[64 * 8];

__global__ __launch_bounds__(256, 1) void attn_bwd_dq64_kernel(
    const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ v, const uint16_t* __restrict__ dout,
    const uint16_t* __restrict__ out_fwd, const float* __restrict__ lse, float* __restrict__ dvec, float* __restrict__ lse2_out,
    const int32_t* __restrict__ cu, int hq, int hkv, float scale, float sl2, int causal, uint16_t* __restrict__ dq, int q_tiles, int n_seq,
    int slot, int xcd_local) {
  __shared__ __attribute__((aligned(1024))) char smem[kLds + 16];
  int& s_item = *reinterpret_cast<int*>(smem + kLds);
  const int n_items = q_tiles * hq * n_seq;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31;
  const int lds_base = (int)lds_u32(smem);
  // (sl2 = scale * log2(e) comes from the host: a float product is a VALU result, and the body wants it in a scalar register)
  const int rep = hq / hkv;
  XcdQueue wq{&g_dq64_work_counter[8 * max(slot, 0)], n_seq * hkv, q_tiles * rep, xcc_id(), 0};
  for (int round = 0;; ++round) {
    int t_end, head, seq;
    if (slot < 0 || xcd_local != 1) {
      if (slot == -1 && round > 0) return;
      const int item = slot == -2 ? static_deal_item(round) : slot < 0 ? (int)blockIdx.x : global_queue_next(wq.counters, n_items, &s_item, tid);
      if (item < 0 || item >= n_items) return;
      t_end = item / (hq * n_seq);                               // query tile counted from the heaviest one
      const int rest = item % (hq * n_seq);
      head = rest % hq, seq = rest / hq;
    } else {
      const int item = xcd_queue_next(wq, &s_item, tid);
      if (item < 0) return;
      const int pair = item / wq.per_pair, r = item % wq.per_pair;
      seq = pair / hkv;
      t_end = r / rep;
      head = (pair % hkv) * rep + r % rep;
    }
    const int qs = cu[seq];
    const int len = cu[seq + 1] - qs;
    // query tiles aligned to the END of the sequence under the causal mask (the partial tile is the cheap first one), as attn_bwd_dq_kernel
    int q0, q_lim;
    auto skip_run = [&]() {      // empty item of the single queue: move the counter past the empty run (attn_common.h)
      if (slot >= 0 && xcd_local == 2 && head == 0)
        queue_skip_empty_run(wq.counters, tid, cu, n_seq, hq, t_end, seq, [&](int level, int ql) {
          return causal ? ql - level * kBlockQ > 0 : (q_tiles - 1 - level) * kBlockQ < ql;
        });
    };
    if (causal) {
      q_lim = len - t_end * kBlockQ;
      if (q_lim <= 0) { skip_run(); continue; }
      q0 = max(0, q_lim - kBlockQ);
    } else {
      q_lim = len;
      q0 = (q_tiles - 1 - t_end) * kBlockQ;
      if (q0 >= len) { skip_run(); continue; }
    }
    const int kvh = head / rep;
    const int wave_qmin = q0 + 64 * wave;
    const int wave_qmax = min(wave_qmin + 63, q_lim - 1);
    const int nvalid = __builtin_amdgcn_readfirstlane(q_lim - wave_qmin);           // rows of this wave that exist (<= 0: none)
    const int my_qa = min(wave_qmin + j, q_lim - 1), my_qb = min(wave_qmin + 32 + j, q_lim - 1);
    const int kv_end = causal ? q_lim : len;
    const int n_tiles = __builtin_amdgcn_readfirstlane((kv_end + kTileK - 1) / kTileK);
    // tiles this WAVE computes (the others it only helps to load), and the first one that needs the mask
    int n_w = nvalid > 0 ? n_tiles : 0;
    if (causal && n_w > 0) n_w = min(n_tiles, wave_qmax / kTileK + 1);
    n_w = __builtin_amdgcn_readfirstlane(n_w);
    int mfirst = len / kTileK;
    if (causal) mfirst = min(mfirst, wave_qmin / kTileK + ((wave_qmin % kTileK) != kTileK - 1 ? 0 : 1));
    mfirst = __builtin_amdgcn_readfirstlane(mfirst);
    const int kmax_a = causal ? min(len - 1, my_qa) : len - 1;
    const int kmax_b = causal ? min(len - 1, my_qb) : len - 1;

    const int row0 = min(wave_qmin, q_lim - 1);                                     // (a wave without rows still forms valid addresses)
    const int64_t ro = ((int64_t)(qs + row0) * hq + head) * kD;
    const void* const qbase = uniform_ptr(q + ro);
    const void* const dobase = uniform_ptr(dout + ro);
    const void* const obase = uniform_ptr(out_fwd + ro);
    const void* const dqbase = uniform_ptr(dq + ro);
    const int64_t fo = (int64_t)(qs + row0) * hq + head;
    const void* const lsebase = uniform_ptr(lse + fo);
    const void* const dvecbase = uniform_ptr(dvec + fo);
    const void* const lse2base = uniform_ptr(lse2_out + fo);
    const void* const kbase = uniform_ptr(k + ((int64_t)qs * hkv + kvh) * kD);
    const void* const vbase = uniform_ptr(v + ((int64_t)qs * hkv + kvh) * kD);
    const int qrs2 = hq * kD * 2, krs2 = hkv * kD * 2, fstride = hq * 4;
    const int len_u = __builtin_amdgcn_readfirstlane(len);
    asm volatile(VSEL_DQ64_ASM_TEXT
                 :
                 : [qbase] "s"(qbase), [dobase] "s"(dobase), [obase] "s"(obase), [dqbase] "s"(dqbase), [lsebase] "s"(lsebase),
                   [dvecbase] "s"(dvecbase), [lse2base] "s"(lse2base), [kbase] "s"(kbase), [vbase] "s"(vbase), [qrs2] "s"(qrs2),
                   [krs2] "s"(krs2), [fstride] "s"(fstride), [ntiles] "s"(n_tiles), [nw] "s"(n_w), [mfirst] "s"(mfirst), [len] "s"(len_u),
                   [nvalid] "s"(nvalid), [sl2] "s"(sl2), [scale] "s"(scale), [wave] "s"(wave), [ldsbase] "s"(lds_base),
                   [kmaxa] "v"(kmax_a), [kmaxb] "v"(kmax_b)
                 : VSEL_DQ64_ASM_CLOBBERS);
    __syncthreads();                   // the next item's first loads overwrite staging rows / ring slots other waves may still read
  }
}

namespace bwd {
// attn_bwd.hip's launcher hands the dQ pass over here for long sequences (knob attn_bwd_dq64): 256-query items
int dq64_launch(hipStream_t st, const void* q, const void* k, const void* v, const void* dout, const void* out, const float* lse,
                float* dvec, float* lse2, const int32_t* cu, int64_t n_seq, int64_t max_seqlen, int64_t hq, int64_t hkv, float scale,
                int causal, void* dq, int xcd_local) {
  const int q_tiles = (int)cdiv(max_seqlen, kBlockQ);
  const int64_t n_items = (int64_t)q_tiles * hq * n_seq;
  if (n_items >= (1ll << 31)) return fail(VSEL_ERR_UNSUPPORTED, "too many attention work items");
  int slot = -1, taken = -1;                      // (a counter slot per queued launch: common.h, queue_slot_acquire)
  if (attn_static_deal(n_items, 256, true, 36)) {
    slot = -2;
  } else if (n_items > 256) {
    if (int rc = queue_slot_acquire(kSlotDq64, st, &taken)) return rc;
    slot = taken;
    int* counters = nullptr;
    VSEL_HIP_CHECK(hipGetSymbolAddress((void**)&counters, HIP_SYMBOL(g_dq64_work_counter)));
    VSEL_HIP_CHECK(hipMemsetAsync(counters + 8 * slot, 0, 8 * sizeof(int), st));
  }
  VSEL_LAUNCH(attn_bwd_dq64_kernel, dim3((unsigned)std::min<int64_t>(n_items, 256)), dim3(256), 0, st, (const uint16_t*)q,
                     (const uint16_t*)k, (const uint16_t*)v, (const uint16_t*)dout, (const uint16_t*)out, lse, dvec, lse2, cu, (int)hq, (int)hkv,
                     scale, scale * 1.4426950408889634f, causal, (uint16_t*)dq, q_tiles, (int)n_seq, slot, xcd_local);
  queue_slot_launched(kSlotDq64, taken, st);
  VSEL_AFTER_LAUNCH(st, "attn_bwd_dq64_kernel");
  return VSEL_OK;
}
}  // namespace bwd

}  // namespace vsel
