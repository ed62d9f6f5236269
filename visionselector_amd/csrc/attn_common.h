// Shared pieces of the attention kernels (attn.hip forward, attn_bwd.hip backward): the LDS tile layout and the per-lane
// fragment addresses of v_mfma_f32_32x32x16_bf16 operands.
//
// A tile is 64 rows x 256 bytes (one row = one key / query, up to 128 bf16 features).  The 16-byte part p of row r lives at
// part p ^ swz(r), swz(r) = ((r & 3) << 2) | ((r >> 2) & 3).  That single layout is conflict-free for
//   * row reads (ds_read_b128): the 16 lanes of a group read 16 rows whose (r & 15) are all different -> 16 different
//     physical parts -> all 64 banks;
//   * transposed reads (ds_read_b64_tr_b16): a 16-lane group reads a [4 rows x 16 columns] block; rows 4a .. 4a+3 move the
//     64-byte column block to 4 different 64-byte groups -> all 64 banks over the two groups served together.
// Tiles are filled by global_load_lds_dwordx4, which writes LDS lane-linearly (wave-uniform base + 16 * lane): the swizzle
// is therefore applied to the SOURCE address -- LDS position (r, p) receives global part p ^ swz(r) (swz is an involution
// on the part index for a fixed row).
#pragma once
#include "common.h"

namespace vsel {
namespace attn {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));

constexpr int kRowBytes = 256;
constexpr int kTileRows = 64;
constexpr int kTileBytes = kTileRows * kRowBytes;   // 16 KiB

__device__ __forceinline__ bf16x8_t as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8_t, v); }
__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int chunk_off(int row, int part) { return row * kRowBytes + ((part ^ swz(row)) << 4); }

// A-operand row i of a 32-row block holds tile row perm_row(i) (bits 2 and 3 swapped), so that C registers 8m .. 8m+7 of
// lane half hh are the 8 CONSECUTIVE tile rows 16m + 8hh .. +7 and feed the next MFMA's B operand without a shuffle.
__device__ __forceinline__ int perm_row(int j) { return (j & 0x13) | ((j & 4) << 1) | ((j & 8) >> 1); }

// Row-form A fragments: lane (j, hh) reads 16 bytes = features 16 st + 8 hh .. +7 of tile row perm_row(j) (+ 32 per block,
// added as an immediate by the caller).
template <int KSTEPS>
__device__ __forceinline__ void make_row_addr(int (&row_addr)[KSTEPS], int j, int hh) {
  const int row = perm_row(j);
#pragma unroll
  for (int st = 0; st < KSTEPS; ++st) row_addr[st] = chunk_off(row, 2 * st + hh);
}

// Transposed A fragments: for d-tile dt, lane (i, hh) receives the 8 consecutive tile rows rbase + 8 hh .. +7 of feature
// 32 dt + i as two ds_read_b64_tr_b16 (rows +0..3 and +4..7); within a 16-lane group lane p supplies the address of row
// p >> 2, 4-feature chunk p & 3.  rbase (a multiple of 16) is added as an immediate by the caller.
template <int DTILES>
__device__ __forceinline__ void make_tr_addr(int (&tr_addr)[DTILES][2], int lane) {
  const int p16 = lane & 15, hh = lane >> 5;
#pragma unroll
  for (int dt = 0; dt < DTILES; ++dt)
#pragma unroll
    for (int hi = 0; hi < 2; ++hi)
      tr_addr[dt][hi] = chunk_off(8 * hh + (p16 >> 2) + 4 * hi, 4 * dt + 2 * ((lane >> 4) & 1) + ((p16 & 3) >> 1)) + 8 * (p16 & 1);
}

__device__ __forceinline__ bf16x8_t tr_read(const char* tile, const int (&addr)[2], int imm) {
  typedef __attribute__((address_space(3))) bf16x4_t* lds_p;
  const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(tile + addr[0] + imm));
  const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(tile + addr[1] + imm));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Source-swizzled direct load of one 1-KiB slice (4 tile rows) of a tile: lane l lands at (row 4 i + (l >> 4), position
// l & 15).  `src_row_ptr` points at the first feature of that lane's source row; `part8` = source part * 8 elements.
__device__ __forceinline__ void load_slice_lds(const uint16_t* src_row_ptr, int part8, char* tile, int i) {
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  __builtin_amdgcn_global_load_lds((gptr_t)(src_row_ptr + part8), (lptr_t)(tile + i * 1024), 16, 0, 0);
}

// Source part (in 16-byte units) for lane l of wave w issuing slices w, w + n_waves, ...: position (l & 15) of row
// 4 i + (l >> 4) holds part (l & 15) ^ swz(row), and swz(row) = ((l >> 4) << 2) | (w & 3) for every such slice.
__device__ __forceinline__ int slice_src_part(int lane, int wave) { return (lane & 15) ^ (((lane >> 4) << 2) | (wave & 3)); }

constexpr int kTileK = 64;           // keys per K / V tile
constexpr float kLazyTau = 8.f;      // online softmax: the reference exponent of a row moves only past this slack (log2 units)

// Optional paged-KV / separate key lengths.  Contiguous var-len prefill (the reference's call sites) leaves it empty.
struct PagedKV {
  const int32_t* seqlens_k;     // [n_seq] key length per sequence (NULL: same cu_seqlens as the queries)
  const int32_t* block_table;   // [n_seq, max_pages] physical page of logical page p (NULL: keys contiguous at cu_k)
  const int32_t* cu_k;          // [n_seq + 1] key row offsets when keys are contiguous but differ from the queries (or NULL)
  int max_pages;
  int page_size;
  // Layout of q and of k / v in elements (0 = packed [T, H, d]: row stride H * d, head stride d).  Head-major tensors
  // ([B, H, L, d] as HuggingFace attention modules hold them): row stride d, head stride L * d; the sequence base stays
  // cu[seq] * H * d in both layouts (cu = b * L).
  int64_t q_row_stride, q_head_stride, kv_row_stride, kv_head_stride, v_row_stride, v_head_stride;   // v_*: 0 = same as k
};

// 64-rows-per-wave forward for head_dim 128 (attn_fwd64.hip): contiguous keys only (no page table); same items as the 8-wave form
int attn_fwd64_launch(hipStream_t st, const void* q, const void* k, const void* v, const int32_t* cu_q, int64_t n_seq,
                      int64_t max_seqlen_q, int64_t hq, int64_t hkv, float scale, int causal, void* out, const PagedKV& pg, float* lse);

// group-shared forward for short sequences (attn_fwd_gqa.hip): one 8-wave workgroup serves a kv head's whole q-head group on one query
// tile; contiguous keys of the queries' own sequences only (prefill), head_dim 128, 2 <= hq / hkv <= 8
int attn_fwd_gqa_launch(hipStream_t st, const void* q, const void* k, const void* v, const int32_t* cu_q, int64_t n_seq,
                        int64_t max_seqlen_q, int64_t hq, int64_t hkv, float scale, int causal, void* out, const PagedKV& pg, float* lse,
                        bool deal);      // deal: items dealt out statically (b, 2 G - 1 - b, 2 G + b, ...) instead of drawn from the queue

// ... and on the generated 64-rows-per-wave loop, two heads per wave (attn_fwd_gqa64.hip)
int attn_fwd_gqa64_launch(hipStream_t st, const void* q, const void* k, const void* v, const int32_t* cu_q, int64_t n_seq,
                          int64_t max_seqlen_q, int64_t hq, int64_t hkv, float scale, int causal, void* out, const PagedKV& pg, float* lse,
                          bool deal);

// key-range parts of attn_fwd64's items for one long sequence / a few of equal length (attn_fwd64_parts.hip; needs a caller workspace)
struct PartsPlan {
  int q_tiles, cap, n_items, max_parts;
  int level_off[65];                    // items in front of tile level l (level 0 = the LAST query tile: the heaviest), [q_tiles] = n_items
};
bool fwd64_parts_plan(int64_t n_seq, int64_t len, int64_t hq, int cap, PartsPlan* plan);
size_t fwd64_parts_workspace_bytes(int64_t n_seq, int64_t len, int64_t hq, int max_parts);
int attn_fwd64_parts_launch(hipStream_t st, const void* q, const void* k, const void* v, int64_t n_seq, int64_t len, int64_t hq, int64_t hkv,
                            float scale, void* out, float* lse, const PartsPlan& plan, void* ws, size_t ws_bytes);

// ---- XCD-local work queues (round 4) ------------------------------------------------------------------------------------------
// Every attention kernel streams one operand pair (K / V in the forward and the dQ pass, Q / dO in the dK / dV pass) that ALL the
// work items of a (sequence, kv head) pair share.  With one global queue the workgroups resident on an XCD belong to 8+ different
// pairs and the shared stream misses that XCD's 4 MB L2 (measured at 16 x 4096: L2 hit rate 19 % in the dK / dV pass, 63 % dQ,
// 69 % forward; 4.6 TB/s of memory-side reads in the dK / dV pass -- it ran at the fabric limit, not at the matrix pipe).
// Pair p is therefore queued on XCD p % 8: a workgroup takes items from the queue of the XCD it runs on (HW_REG_XCC_ID) and, when
// that queue is empty, helps the next one (p % 8 + 1, ...), so ragged batches still balance.  Placement only changes speed: which
// workgroup computes an item never changes the item's arithmetic (outputs stay bit-identical).
//   counters: int[8], zeroed by the host before the launch.  -> item index in [0, n_pairs * per_pair), or -1 when all is done.
//   item index = pair * per_pair + r, r = the pair's items heaviest first.
struct XcdQueue {
  int* counters;
  int n_pairs, per_pair;
  int cur, tried;            // queue being drained, queues found empty so far
  int couples;               // != 0: pairs are queued in COUPLES (2 c, 2 c + 1), couple c on XCD c % 8 (the dK / dV pass: dkdv_walks_up)
};
// dK / dV pass (attn_bwd.hip, attn_bwd_dkdv64.hip): the items of a (sequence, kv head) pair walk their query tiles from the sequence's END
// downward when the kv head is even and from the item's first query UPWARD when it is odd (a rule of the item alone: a sequence's dK / dV
// do not depend on the batch around it).  With couples on one XCD queue the two directions alternate there: a workgroup that finishes key
// block j of the downward pair after 2 (n - j) tile steps draws key block n - 1 - j ... of the upward pair and starts at the tile that
// pair's sweep has reached by then, so the CUs of an XCD stay on ONE Q / dO tile per pair instead of fanning out behind each other.
// Measured (16 sequences, 28 / 4 heads, dK / dV kernel): 4096 tokens -2.5 % (L2 hit 0.63 -> 0.94, memory-side reads 65 M -> 9 M: a pair's 32
// key blocks = the 32 CUs of an XCD), 3072 ... 6144 tokens -0.2 ... -1.1 %, 8192 tokens +0.5 % (64 key blocks: the second cohort of a pair
// starts at the end again) -- hence only sequences of up to 6144 tokens turn around.
__device__ __forceinline__ int dkdv_walks_up(int kv_head, int updown, int len) {
  return updown != 0 && (kv_head & 1) && len <= 6144 ? 1 : 0;
}
// one global queue in the legacy item order (short sequences: a pair's stream is small and heaviest-first over ALL pairs balances
// better than per-XCD lists): -> item in [0, n_items) or -1
__device__ __forceinline__ int global_queue_next(int* counter, int n_items, int* s_item, int tid) {
  if (tid == 0) {
    const int idx = atomicAdd(counter, 1);
    *s_item = idx < n_items ? idx : -1;
  }
  __syncthreads();
  const int item = *s_item;
  __syncthreads();
  return __builtin_amdgcn_readfirstlane(item);
}
// static deal (slot == -2; few rounds of items, knob attn_static): workgroup b takes items b, 2 G - 1 - b, 2 G + b, ... of the
// heaviest-first list -- no atomic round trip and no hand-over barriers in front of an item
__device__ __forceinline__ int static_deal_item(int round) {
  const int g = (int)gridDim.x, b = (int)blockIdx.x;
  return round * g + ((round & 1) ? g - 1 - b : b);
}
// The single queue on RAGGED batches.  The item list is (level, sequence, head) with `width` heads per (level, sequence) and as many
// levels as the LONGEST sequence has tiles, so a shorter sequence owns empty items at the levels it does not reach -- 40 % of the list
// for prompts of 131 ... 947 tokens -- and every empty item costs the workgroup that draws it an atomic round trip, two barriers and the
// cu_seqlens loads.  A workgroup that has drawn an empty item therefore moves the shared counter past the whole empty RUN before it draws
// again: the 64 lanes of wave 0 look at the following sequences of that level (ballot) and one atomicMax sets the counter to the first item
// with work (only empty items are ever skipped: everything below the counter's old value was taken, everything from the drawn item up to
// the new value is empty).  `nonempty(level, qlen)` is the kernel's own rule.  Placement only.  (Checking BEFORE handing an item over --
// the cu loads in front of every draw's barrier -- was measured first: -3 ... -9 % on prompts of 131 ... 947 tokens, the extra round trip
// sits on every item's critical path.)
// Lane 0's atomicMax is ONE asm statement with EXEC narrowed inside: under `if (tid == 0)` in a loop the structurizer let lane 0 leave
// for its atomic while lanes 1 .. 63 ran on without it -- ballot / readfirstlane then saw a wave without lane 0 and never terminated;
// with per-lane neutral operands the atomic optimizer emits a 64-step readlane loop.
__device__ __forceinline__ void lane0_atomic_max(int* p, int v) {
  uint64_t save;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_smax %1, %2, off\n\ts_mov_b64 exec, %0" : "=&s"(save) : "v"(p), "v"(v) : "memory");
}
// CALLERS: every thread of the workgroup calls it (workgroups of >= 64 threads; the single-wave PACK form never does); the work is done
// by wave 0 with all 64 lanes active -- lane0_atomic_max narrows EXEC by hand and the ballot needs the whole wave.
template <class NonEmpty>
__device__ __forceinline__ void queue_skip_empty_run(int* counter, int tid, const int32_t* __restrict__ cu, int n_seq, int width, int level,
                                                     int seq, NonEmpty nonempty) {
  if (tid < 64) {
    int s0 = seq + 1;
    unsigned long long m;
    for (;;) {
      const int s = s0 + tid;
      const bool ne = s < n_seq && nonempty(level, cu[s + 1] - cu[s]);
      m = __ballot(ne);
      if (m != 0 || s0 + 64 >= n_seq) break;
      s0 += 64;
    }
    lane0_atomic_max(counter, m != 0 ? (level * n_seq + s0 + (int)__builtin_ctzll(m)) * width : (level + 1) * n_seq * width);
  }
}
__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
  return v & 7;
}
// thread 0 of the workgroup fetches, the result travels through *s_item (LDS); two barriers per item, as the single queue had
__device__ __forceinline__ int xcd_queue_next(XcdQueue& q, int* s_item, int tid) {
  if (tid == 0) {
    int item = -1;
    while (q.tried < 8) {
      int mine;                                                                   // pairs on queue cur
      if (q.couples) {
        const int n_couples = (q.n_pairs + 1) >> 1;
        const int mc = q.cur < n_couples ? (n_couples - q.cur + 7) >> 3 : 0;      // couples cur, cur + 8, ...
        mine = 2 * mc - ((mc > 0 && (q.n_pairs & 1) && ((n_couples - 1) & 7) == q.cur) ? 1 : 0);   // (the last couple may be a single pair)
      } else {
        mine = q.cur < q.n_pairs ? (q.n_pairs - q.cur + 7) >> 3 : 0;              // pairs cur, cur + 8, ...
      }
      const int idx = mine > 0 ? atomicAdd(&q.counters[q.cur], 1) : 0;
      if (idx < mine * q.per_pair) {
        const int jp = idx / q.per_pair;                                          // position of the pair on this queue
        const int pair = q.couples ? 16 * (jp >> 1) + 2 * q.cur + (jp & 1) : q.cur + 8 * jp;
        item = pair * q.per_pair + idx % q.per_pair;
        break;
      }
      q.cur = (q.cur + 1) & 7;
      ++q.tried;
    }
    *s_item = item;
  }
  __syncthreads();
  const int item = *s_item;
  __syncthreads();
  return __builtin_amdgcn_readfirstlane(item);
}

// ---- LDS reads the compiler does not schedule ----------------------------------------------------------------------------
// hipcc turns the tile body into "ds_read; s_waitcnt lgkmcnt(0); v_mfma" triples (every MFMA behind a full LDS round trip,
// MFMA-busy 34 %) and, for the transpose-read builtin, puts s_waitcnt vmcnt(0) in front of the first V read, which drains the
// direct-to-LDS prefetch of the next tile.  The reads are therefore issued from inline asm in BATCHES (8 per batch, the next
// batch in flight behind the MFMAs of the current one) with counted lgkmcnt waits that name the batch's registers
// (cdna_hip_programming.md 5.7, form (ii)) followed by sched_barrier(0) so that no MFMA is hoisted above its wait (rule 18).
// The asm outputs are "ready" for the compiler at once although the data lands only at the lds_wait* that names them: a
// compiler-inserted COPY or SPILL of such a register between issue and wait would read a stale value.  Guards: the outputs are
// early-clobber (never share a register with the address), every user kernel must compile with 0 spilled VGPRs -- checked at
// build time by visionselector_amd/build.py (-Rpass-analysis=kernel-resource-usage on attn.hip and attn_bwd.hip, kernels listed
// in build.ASM_READ_KERNELS: 0 spilled VGPRs and 0 bytes of scratch or the build fails) --
// and VSEL_HIPCC_FLAGS refuses optimisation-level / debug flags that would change the register allocation wholesale.
__device__ __forceinline__ uint32_t lds_u32(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
// per-lane base address in a VGPR + a compile-time immediate (buffer, key block): no address arithmetic in the tile loop and no
// address registers beyond the 8 + 8 per-lane bases
template <int OFF>
__device__ __forceinline__ u32x4 lds_read_b128_asm(uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
  u32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(r) : "v"(addr), "n"(OFF));
  return r;
}
template <int OFF>
__device__ __forceinline__ u32x2 lds_read_tr16_b64_asm(uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(r) : "v"(addr), "n"(OFF));
  return r;
}
// wait until at most N LDS operations of this wave are outstanding; the 8 registers of the batch that must have landed are
// read-write operands, so their consumers cannot be scheduled above the wait and their values stay in these registers
template <int N>
__device__ __forceinline__ void lds_wait8(u32x4 (&a)[8]) {
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
               : "n"(N));
  __builtin_amdgcn_sched_barrier(0);
}
template <int N>
__device__ __forceinline__ void lds_wait4(u32x4 (&a)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "n"(N));
  __builtin_amdgcn_sched_barrier(0);
}
template <int N>
__device__ __forceinline__ void lds_wait4(u32x2 (&a)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "n"(N));
  __builtin_amdgcn_sched_barrier(0);
}
template <int N>
__device__ __forceinline__ void lds_wait8(u32x2 (&a)[8]) {
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
               : "n"(N));
  __builtin_amdgcn_sched_barrier(0);
}


}  // namespace attn
}  // namespace vsel
