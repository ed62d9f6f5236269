// dK / dV pass of the var-len causal / full GQA attention backward, head_dim 128, bf16 -- one 512-register wave per SIMD with the MFMA
// stream software-pipelined inside the wave (gfx950).
//
// Work split and arithmetic of attn_bwd_dkdv_kernel<SPLIT> (attn_bwd.hip; reference: the autograd of the eager formula,
// qwen-evaluation/qwen25vl/modeling_qwen2_5_vl.py:777-797, reached in training through qwen-vl-finetune/qwenvl/train/trainer.py:101-113):
// item = (128-key block, kv head, sequence); wave w owns keys 32 w .. 32 w + 31 (K / V fragments in registers) and loops over the q heads
// of the group and the 64-query Q / dO tiles, which stream through three-slot direct-to-LDS rings; dK / dV are bit-identical to that
// kernel's (same MFMAs per accumulator in the same order).  Split form (few items): item = (key block, Q head, sequence), raw fp32
// partial rows for attn_bwd_group_sum_kernel, as attn_bwd_dkdv_kernel<true>.  What is new is the schedule: units of (tile, 32-query block), a unit's
// exponentials and dS in the gaps of the OTHER query block's 32 MFMAs, transposed fragments and per-query lse2 / D read ahead -- as one
// GENERATED inline-asm statement per item (tools/gen_attn_bwd_dkdv64.py; hipcc cannot hold the register plan: attn_fwd64.hip).
#include "attn_common.h"
#include <atomic>
#ifndef VSEL_DKDV64_BODY
#define VSEL_DKDV64_BODY "attn_bwd_dkdv64_body.inc"
#endif
#include VSEL_DKDV64_BODY

#include <algorithm>

namespace vsel {

using namespace attn;

namespace {
__device__ __forceinline__ const void* uniform_ptr(const void* p) {
  const uint64_t u = (uint64_t)(uintptr_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return (const void*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
constexpr int kD = 128;
}  // namespace

__device__ int g_dkdv64_work_counter[64 * 8];

__global__ __launch_bounds__(256, 1) void attn_bwd_dkdv64_kernel(
    const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ v, const uint16_t* __restrict__ dout,
    const float* __restrict__ lse2, const float* __restrict__ dvec, const int32_t* __restrict__ cu, int hq, int hkv, float scale, float sl2,
    int causal, uint16_t* __restrict__ dk, uint16_t* __restrict__ dv, float* __restrict__ dk_part, float* __restrict__ dv_part, int split,
    int k_blocks, int n_seq, int slot, int xcd_local_arg) {
  const int updown = xcd_local_arg >> 4, xcd_local = xcd_local_arg & 15;      // (bit 4: attn_common.h, dkdv_walks_up)
  __shared__ __attribute__((aligned(1024))) char smem[VSEL_DKDV64_LDS_BYTES + 16];
  int& s_item = *reinterpret_cast<int*>(smem + VSEL_DKDV64_LDS_BYTES);
  // split = q heads per item (attn_bwd.hip bwd_split_heads; 0: the whole group inside the item).  1 (few items): item = (key block, Q head,
  // sequence) writing its fp32 partial [T, hq, 128], attn_bwd_group_sum_kernel adds the heads of a group -- the item numbering of
  // attn_bwd_dkdv_kernel<SPLIT>.  k > 1 (an unsplit grid of 1 - 2 rounds, where the heaviest item is the run time): a group's heads in
  // ceil(rep / k) PARTS, each part one item that loops over its heads and leaves its fp32 partial in the rows of its FIRST head; the group sum
  // adds the parts' rows in ascending order.
  const int parts = split ? (hq / hkv + split - 1) / split : 1;
  const int heads_dim = hkv * parts;
  const int n_items = k_blocks * heads_dim * n_seq;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hh = lane >> 5;
  const int lds_base = (int)lds_u32(smem);
  const int rep = hq / hkv;
  XcdQueue wq{&g_dkdv64_work_counter[8 * max(slot, 0)], n_seq * hkv, k_blocks * parts, xcc_id(), 0, updown};
  for (int round = 0;; ++round) {
    int kblock, hsel, seq;
    if (slot < 0 || xcd_local != 1) {
      if (slot == -1 && round > 0) return;
      const int item = slot == -2 ? static_deal_item(round) : slot < 0 ? (int)blockIdx.x : global_queue_next(wq.counters, n_items, &s_item, tid);
      if (item < 0 || item >= n_items) return;
      kblock = item / (heads_dim * n_seq);
      const int rest = item % (heads_dim * n_seq);
      hsel = rest % heads_dim, seq = rest / heads_dim;
    } else {
      const int item = xcd_queue_next(wq, &s_item, tid);
      if (item < 0) return;
      const int pair = item / wq.per_pair, r = item % wq.per_pair;
      seq = pair / hkv;
      kblock = r % k_blocks;
      hsel = (pair % hkv) * parts + r / k_blocks;
    }
    const int kvh = hsel / parts;
    const int first_head = kvh * rep + (hsel % parts) * split;                                  // (split == 0: parts == 1, the group's first head)
    const int n_heads = split ? min(split, rep - (hsel % parts) * split) : rep;
    const int qs = cu[seq];
    const int len = cu[seq + 1] - qs;
    const int k0 = kblock * 128;
    if (k0 >= len) {      // empty item of the single queue: move the counter past the empty run (attn_common.h), once per (level, sequence) group
      if (slot >= 0 && xcd_local == 2 && hsel == 0)
        queue_skip_empty_run(wq.counters, tid, cu, n_seq, heads_dim, kblock, seq, [&](int level, int ql) { return level * 128 < ql; });
      continue;
    }
    const int kw0 = __builtin_amdgcn_readfirstlane(k0 + 32 * wave);
    const int my_k = min(kw0 + j, len - 1);
    const int kvalid = (kw0 + j) < len ? 1 : 0;
    const int q_begin = causal ? k0 : 0;                         // a multiple of 128, hence of the 64-query tile
    const int tiles_per_head = (len - q_begin + kTileK - 1) / kTileK;
    const int n_iter = __builtin_amdgcn_readfirstlane(tiles_per_head * n_heads);
    const int head0 = __builtin_amdgcn_readfirstlane(first_head);
    const int nheads_u = __builtin_amdgcn_readfirstlane(n_heads);
    const int dirup = __builtin_amdgcn_readfirstlane(dkdv_walks_up(kvh, updown, len));

    const void* const qbase = uniform_ptr(q + (int64_t)qs * hq * kD);
    const void* const dobase = uniform_ptr(dout + (int64_t)qs * hq * kD);
    const void* const lsebase = uniform_ptr(lse2 + (int64_t)qs * hq);
    const void* const dbase = uniform_ptr(dvec + (int64_t)qs * hq);
    const int64_t ro = ((int64_t)(qs + my_k) * hkv + kvh) * kD;
    const uint16_t* const kptr = k + ro + 8 * hh;
    const uint16_t* const vptr = v + ro + 8 * hh;
    const int64_t po = ((int64_t)(qs + my_k) * hq + first_head) * kD + 4 * hh;
    void* const dkptr = split ? (void*)(dk_part + po) : (void*)(dk + ro + 4 * hh);
    void* const dvptr = split ? (void*)(dv_part + po) : (void*)(dv + ro + 4 * hh);
    const int split_u = __builtin_amdgcn_readfirstlane(split != 0 ? 1 : 0);
    const int qrs2 = hq * kD * 2, fstride = hq * 4;
    const int len_u = __builtin_amdgcn_readfirstlane(len), qbeg_u = __builtin_amdgcn_readfirstlane(q_begin);
    asm volatile(VSEL_DKDV64_ASM_TEXT
                 :
                 : [qbase] "s"(qbase), [dobase] "s"(dobase), [lsebase] "s"(lsebase), [dbase] "s"(dbase), [qrs2] "s"(qrs2),
                   [fstride] "s"(fstride), [niter] "s"(n_iter), [qbegin] "s"(qbeg_u), [len] "s"(len_u), [sl2] "s"(sl2), [scale] "s"(scale),
                   [kw0] "s"(kw0), [causal] "s"(causal), [wave] "s"(wave), [head0] "s"(head0), [nheads] "s"(nheads_u), [dirup] "s"(dirup), [ldsbase] "s"(lds_base), [split] "s"(split_u),
                   [kptr] "v"(kptr), [vptr] "v"(vptr), [dkptr] "v"(dkptr), [dvptr] "v"(dvptr), [kvalid] "v"(kvalid)
                 : VSEL_DKDV64_ASM_CLOBBERS);
    __syncthreads();                   // the next item's first loads overwrite ring slots other waves may still read
  }
}

namespace bwd {
// attn_bwd.hip's launcher hands the dK / dV pass over here (knob attn_bwd_dkdv64) when the group's q heads are looped inside an item
int dkdv64_launch(hipStream_t st, const void* q, const void* k, const void* v, const void* dout, const float* lse2, const float* dvec,
                  const int32_t* cu, int64_t n_seq, int64_t max_seqlen, int64_t hq, int64_t hkv, float scale, int causal, void* dk, void* dv,
                  float* dk_part, float* dv_part, int split, int xcd_local) {
  // split = q heads per item (0: the group inside the item); the caller runs attn_bwd_group_sum_kernel behind a split launch
  const int k_blocks = (int)cdiv(max_seqlen, 128);
  const int64_t n_items = (int64_t)k_blocks * hkv * (split ? cdiv(hq / hkv, split) : 1) * n_seq;
  if (n_items >= (1ll << 31)) return fail(VSEL_ERR_UNSUPPORTED, "too many attention work items");
  int slot = -1, taken = -1;                      // (a counter slot per queued launch: common.h, queue_slot_acquire)
  if (attn_static_deal(n_items, 256, true, 36)) {
    slot = -2;
  } else if (n_items > 256) {
    if (int rc = queue_slot_acquire(kSlotDkdv64, st, &taken)) return rc;
    slot = taken;
    int* counters = nullptr;
    VSEL_HIP_CHECK(hipGetSymbolAddress((void**)&counters, HIP_SYMBOL(g_dkdv64_work_counter)));
    VSEL_HIP_CHECK(hipMemsetAsync(counters + 8 * slot, 0, 8 * sizeof(int), st));
  }
  VSEL_LAUNCH(attn_bwd_dkdv64_kernel, dim3((unsigned)std::min<int64_t>(n_items, 256)), dim3(256), 0, st, (const uint16_t*)q,
                     (const uint16_t*)k, (const uint16_t*)v, (const uint16_t*)dout, lse2, dvec, cu, (int)hq, (int)hkv, scale,
                     scale * 1.4426950408889634f, causal, (uint16_t*)dk, (uint16_t*)dv, dk_part, dv_part, split, k_blocks, (int)n_seq, slot,
                     xcd_local);
  queue_slot_launched(kSlotDkdv64, taken, st);
  VSEL_AFTER_LAUNCH(st, "attn_bwd_dkdv64_kernel");
  return VSEL_OK;
}
}  // namespace bwd

}  // namespace vsel
