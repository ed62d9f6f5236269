// Var-len (causal or full) GQA attention forward (gfx950, bf16): head_dim 128 for the compressed-sequence prefill of the LLMs,
// head_dim 80 / 64 for the var-len window attention of the vision towers (Qwen2.5-VL ViT: 1280 / 16, Rice ViT: 1024 / 16).
//
// Replaces flash_attn_varlen_func / the FA2 prefill the reference reaches through
//   qwen-vl-finetune/qwenvl/train/trainer.py:101-113          (cu_seqlens passed as `attention_mask`)
//   qwen-evaluation/qwen25vl/modeling_qwen2_5_vl.py:900       (Qwen2_5_VLFlashAttention2.forward)
//   llava-ov-15/llavaonevision1_5/modeling_llavaonevision1_5.py:686
// Math = the in-tree eager path (modeling_qwen2_5_vl.py:777-797): softmax(q k^T / sqrt(d) + causal) v, fp32 softmax,
// GQA by head // (Hq / Hkv) (repeat_kv, :693-702).  q/k arrive already rotated (M-RoPE / 1-D RoPE upstream).
//
// Structure: one workgroup = 4 waves = 128 queries of one (sequence, q-head); a wave owns 32 queries.
//   S^T = K Q^T   (v_mfma_f32_32x32x16_bf16, A = K tile from LDS, B = Q^T held in registers) so that one lane holds
//                 one query's scores -> row max / sum need a single cross-half exchange (lane ^ 32)
//   O^T += V^T P^T (A = V^T fragment via ds_read_b64_tr_b16 from the row-major V tile, B = P^T packed to bf16 in
//                 registers; the K rows are permuted (bits 2<->3) so each lane's P registers are 8 consecutive keys)
//   K/V tiles of 64 keys are double-buffered in LDS and arrive by global_load_lds_dwordx4 (global -> LDS, no staging
//   registers, no ds_write pass) issued for tile t+1 before the math of tile t.  Both tiles use 256-byte rows with the
//   16-byte part p of row r stored at p ^ swz(r), swz(r) = ((r & 3) << 2) | ((r >> 2) & 3): conflict-free for the K row
//   reads (ds_read_b128) and for the V transpose reads (ds_read_b64_tr_b16); the direct loads write LDS lane-linearly,
//   so the swizzle is applied to their SOURCE address.  Online softmax in exp2 domain, fp32.
#include "attn_common.h"
#include <atomic>

#include <algorithm>
#include <type_traits>

namespace vsel {

using namespace attn;      // tile layout + fragment addressing shared with the backward (attn_common.h)

// LDS rows are 256 bytes for every supported head_dim (64 / 80 / 128)
constexpr int kBuf = kTileBytes;             // 16 KiB per tile
constexpr int kLds = 4 * kBuf;               // K[2], V[2]: 64 KiB

__device__ __forceinline__ bf16x8_t to_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8_t, v); }


// Persistent work-stealing grid: items = (q-tile, head, sequence), handed out heaviest-first (largest q-tile = most KV
// tiles under the causal mask) through one atomic counter, so the causal triangle is load-balanced over the 2 x 256
// resident workgroups instead of being bounded by the last q-tile (1.9x fewer tiles on the critical path at L = 2368).
__device__ int g_attn_work_counter[64 * 8];   // 64 launch slots x 8 XCD-local queues (attn_common.h, XcdQueue)

#ifdef VSEL_TRACE
// s_memtime (shader clock) stamps of ONE steady-state tile of workgroup 0's first item, one row per wave (tools/trace_attn_fwd.py)
__device__ unsigned long long g_fwd_tile_trace[8][8];
#define VSEL_FWD_STAMP(slot)                                                                        \
  do {                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    if (trace_on && lane == 0) g_fwd_tile_trace[wave_all][slot] = __builtin_readcyclecounter();      \
    __builtin_amdgcn_sched_barrier(0);                                                              \
  } while (0)
#else
#define VSEL_FWD_STAMP(slot) do {} while (0)
#endif


// NW = waves per workgroup (4 -> 128 queries, two workgroups per CU; 8 -> 256 queries, one workgroup per CU sharing ONE K/V
// stream among its 8 waves: half the global loads and LDS stores per FLOP, used when the grid is large enough).
// D = head_dim (64, 80, 128): D / 16 k-steps for S, ceil(D / 32) d-tiles for O (columns past D are computed on whatever the
// unused LDS parts hold and never stored).
// PACK (decode / short query chunks against a cache, NW = 1): one wave serves a whole GQA group of one sequence -- lane j is
// the pair (query j / rep, head kvh * rep + j % rep), qlen * rep <= 32 -- so the K/V of a kv head is streamed ONCE for its rep
// query heads instead of once per head, and a workgroup is a single wave (4x more sequences resident).
// KVS = 2 (NW = 8, latency-bound small grids): the workgroup's two 4-wave groups serve the SAME 128 queries and split the KV
// tiles between them (group g takes tiles g, g + 2, ...; each group streams its own K/V tiles through its own LDS buffers),
// then merge their online-softmax states through LDS -- the dependent chain of KV tiles, which is what a single short
// sequence costs, is halved (L' = 524: 9 tiles -> 5).
// KH = 2 (with KVS = 2, one short sequence): 64 queries per workgroup, and the two waves that share a 32-query slice inside a
// KV stream take one 32-key block of every tile each (four partial softmax states per slice, merged at the end).
// tools/trace_attn.py on L' = 524: with 128-query workgroups a round costs 2.35 us -- two waves per SIMD run the same phase at
// the same time and their MFMA and VALU time add up -- while 116 of the 256 CUs have no workgroup (140 workgroups); key halves
// halve the work of a wave per round and 64-query workgroups use 252 CUs.
// launch bounds: two waves per SIMD for the 4- / 8-wave workgroups; the single-wave GQA-packed decode workgroup owns 32 KiB of
// LDS by itself, so waves-per-SIMD is LDS-bound there (5 workgroups per CU ~ 1 per SIMD) and asking for 2 only drew a warning
template <bool USE_TR, int NW, int D, bool PACK = false, int KVS = 1, int KH = 1, bool TAIL = false>
__global__ __launch_bounds__(64 * NW, NW >= 4 ? 2 : 1) void varlen_attn_fwd_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                              const uint16_t* __restrict__ v,
                                                              const int32_t* __restrict__ cu, int hq, int hkv,
                                                              float scale_log2e, int causal, uint16_t* __restrict__ out,
                                                              int q_tiles, int n_seq, int slot, PagedKV pg,
                                                              float* __restrict__ lse) {
  static_assert(!TAIL || (!PACK && KVS == 1), "end-aligned query tiles: single-stream per-head form only");
  constexpr int GW = NW / KVS;                     // waves that share one K/V stream
  constexpr int QW = GW / KH;                      // ... of which QW own distinct 32-query slices (KH waves per slice)
  constexpr int kBlockQ = 32 * QW;
  constexpr int kLoadsPerWave = 16 / GW;           // 1-KiB wave-instructions per tile per tensor
  constexpr int NKB = 2 / KH;                      // 32-key blocks of a tile that one wave computes
  static_assert(KVS == 1 || (KVS == 2 && NW == 8 && !PACK), "two KV streams need 8 waves");
  static_assert(KH == 1 || (KH == 2 && KVS == 2 && USE_TR && D == 128), "key halves: only in the two-stream form");
  constexpr int kHeadDim = D;
  constexpr int kSteps = D / 16;                   // k-steps of S^T = K Q^T
  constexpr int kDTiles = (D + 31) / 32;           // 32-wide d-tiles of O^T
  constexpr int kParts = D / 8;                    // 16-byte parts per row that hold data
  // ONE __shared__ object: with a second one (a 4-byte work-item slot) hipcc puts s_waitcnt vmcnt(0) in front of the first ds_read
  // of every tile, which drains the direct-to-LDS prefetch of the NEXT tile before this tile's math (cdna_hip_programming.md,
  // "Three .s-level traps", (a))
  __shared__ __attribute__((aligned(16))) char smem[KVS * kLds + 16];
  int& s_item = *reinterpret_cast<int*>(smem + KVS * kLds);
  static_assert(!PACK || NW == 1, "the GQA-packed form is single-wave");
  const int rep = hq / hkv;
  const int n_items = PACK ? hkv * n_seq : q_tiles * hq * n_seq;
  const int64_t q_rs = pg.q_row_stride ? pg.q_row_stride : (int64_t)hq * kHeadDim;
  const int64_t q_hs = pg.q_row_stride ? pg.q_head_stride : kHeadDim;
  const int64_t kv_rs = pg.kv_row_stride ? pg.kv_row_stride : (int64_t)hkv * kHeadDim;
  const int64_t kv_hs = pg.kv_row_stride ? pg.kv_head_stride : kHeadDim;
  const int64_t v_rs = pg.v_row_stride ? pg.v_row_stride : kv_rs;
  const int64_t v_hs = pg.v_row_stride ? pg.v_head_stride : kv_hs;
  const int tid = threadIdx.x, lane = tid & 63, wave_all = tid >> 6;
  const int grp = KVS == 1 ? 0 : __builtin_amdgcn_readfirstlane(wave_all / GW);      // KV stream of this wave (uniform)
  const int wave = KVS == 1 ? wave_all : wave_all % GW;        // wave inside its group (its share of the tile loads)
  const int qw = wave % QW;                                    // its 32-query slice
  const int kh = KH == 1 ? 0 : __builtin_amdgcn_readfirstlane(wave / QW);   // its 32-key block of every tile (KH = 2)
  const int pidx = grp * KH + kh;                              // which partial softmax state of the slice it carries
  const int j = lane & 31, hh = lane >> 5;
  char* const k_sm = smem + grp * kLds;
  char* const v_sm = k_sm + 2 * kBuf;
  // A-row i of the K operand holds key pi(i) (bits 2 and 3 swapped) so that C registers 8m..8m+7 of lane half hh are
  // the 8 consecutive keys 16m + 8hh .. +7 of the 32-key block.
  // per-lane LDS byte offsets inside a tile; the buffer, the 32-key block and the 16-key step add immediates.  K rows are
  // read permuted (perm_row) so that one lane's P registers are 8 consecutive keys.
  int row_addr[kSteps], tr_addr[kDTiles][2];
  make_row_addr<kSteps>(row_addr, j, hh);
  make_tr_addr<kDTiles>(tr_addr, lane);
  if constexpr (KVS > 1) {                                     // this group's tile buffers
#pragma unroll
    for (int st = 0; st < kSteps; ++st) row_addr[st] += grp * kLds;
#pragma unroll
    for (int dt = 0; dt < kDTiles; ++dt) { tr_addr[dt][0] += grp * kLds; tr_addr[dt][1] += grp * kLds; }
  }
  // LDS byte addresses of the per-lane bases (for the inline-asm reads)
  uint32_t row_addr_u[kSteps], tr_addr_u[kDTiles][2];
#pragma unroll
  for (int st = 0; st < kSteps; ++st) row_addr_u[st] = lds_u32(smem) + row_addr[st];
#pragma unroll
  for (int dt = 0; dt < kDTiles; ++dt) { tr_addr_u[dt][0] = lds_u32(smem) + tr_addr[dt][0]; tr_addr_u[dt][1] = lds_u32(smem) + tr_addr[dt][1]; }
  if constexpr (KH == 2) {                                     // key halves: the wave reads only tile rows 32 kh .. 32 kh + 31
    const uint32_t blk = (uint32_t)(wave / QW) * 32u * kRowBytes;
#pragma unroll
    for (int st = 0; st < kSteps; ++st) row_addr_u[st] += blk;
#pragma unroll
    for (int dt = 0; dt < kDTiles; ++dt) { tr_addr_u[dt][0] += blk; tr_addr_u[dt][1] += blk; }
  }
  // direct-to-LDS loads: wave w issues wave-instructions w, w + NW, ...; instruction i covers tile rows 4i .. 4i+3, lane l
  // lands at (row 4i + (l >> 4), position l & 15) and therefore fetches global part (l & 15) ^ swz(row)
  // (a source part past the row's data, head_dim < 128, is redirected to part 0: its LDS position is never read for S and only
  // feeds output columns >= D)
  // slice i holds tile rows 4i .. 4i+3, whose swizzle key is ((l >> 4) << 2) | (i & 3); i & 3 == wave & 3 when NW is a
  // multiple of 4, otherwise (single-wave workgroups) it changes from slice to slice
  auto src_part8 = [&](int i) {
    const int src = slice_src_part(lane, GW % 4 == 0 ? wave : i);
    return (src < kParts ? src : 0) * 8;
  };

  // slot: -1 = one item per workgroup; else launch slot | 0x100 when the items sit on XCD-local queues (a (sequence, kv head) pair's
  // query tiles, heaviest first, the group's q heads inner -- attn_common.h)
  const bool xcd_local = !PACK && slot >= 0 && (slot & 0x100) != 0;
  XcdQueue wq{&g_attn_work_counter[8 * (max(slot, 0) & 0xff)], n_seq * hkv, q_tiles * rep, xcc_id(), 0};
  for (int round = 0;; ++round) {
  int item;
  if (slot == -2) {                     // static deal of the heaviest-first list (attn_common.h): packed batches with few rounds of items
    item = static_deal_item(round);
    if (item >= n_items) return;
  } else if (slot < 0) {                // one item per workgroup (n_items <= resident slots): no counter needed
    if (round > 0) return;
    item = blockIdx.x;
    if (item >= n_items) return;
  } else if (xcd_local) {
    item = xcd_queue_next(wq, &s_item, tid);
  } else {
    item = global_queue_next(wq.counters, n_items, &s_item, tid);
  }
  item = __builtin_amdgcn_readfirstlane(item);      // (uniform by construction; lets the address arithmetic run on the scalar ALU)
  if (item < 0) return;
  VSEL_STAMP(0, 0);
  int qtile, head, seq, kvh;
  if constexpr (PACK) {
    qtile = 0;
    kvh = item % hkv;
    seq = item / hkv;
    head = kvh * rep + (j % rep);                             // per lane
  } else if (xcd_local) {
    const int pair = item / wq.per_pair, r = item % wq.per_pair;
    seq = pair / hkv;
    kvh = pair % hkv;
    qtile = r / rep;
    head = kvh * rep + r % rep;
  } else {
    qtile = item / (hq * n_seq);                               // counted from the END of the sequence: 0 = its last (heaviest) tile
    const int rest = item % (hq * n_seq);
    head = rest % hq;
    seq = rest / hq;
    kvh = head / rep;
  }
  const int qs = cu[seq];
  const int qlen = cu[seq + 1] - qs;
  // qtile counts from the END of the longest sequence (0 = heaviest).  Two tilings, bit-identical outputs (key tiles stay aligned
  // to key 0 and a query's arithmetic does not depend on its tile):
  //   start-aligned: tile t covers [t * kBlockQ, (t + 1) * kBlockQ); the PARTIAL tile of a sequence is its last one, i.e. the one
  //     with the most key tiles under the causal mask (L / 64 rounds for a handful of queries);
  //   TAIL: tiles aligned to the END of the sequence, [qlen - (t + 1) * kBlockQ, qlen - t * kBlockQ): the partial tile is the
  //     first one (one or two key tiles), at the price of a diagonal that crosses one more key tile in every other tile -- about
  //     one tile-round per query tile less in all.  Pays when the grid is throughput-bound (many items per workgroup: packed
  //     batches of compressed sequences), not when one sequence's longest chain is the run time.
  int q0, q_lim;                        // this tile's queries: [q0, min(q0 + kBlockQ, q_lim))
  // an empty item of the single queue (a level this sequence does not reach): move the shared counter past the whole empty run
  auto skip_run = [&]() {
    if constexpr (!PACK)
      if (slot >= 0 && (slot & 0x200) && head == 0)            // (one scan per empty (level, sequence) group, by whoever drew its first head)
        queue_skip_empty_run(wq.counters, tid, cu, n_seq, hq, qtile, seq, [&](int level, int ql) {
          return TAIL ? ql - level * kBlockQ > 0 : (q_tiles - 1 - level) * kBlockQ < ql;
        });
  };
  if constexpr (TAIL) {
    q_lim = qlen - qtile * kBlockQ;
    if (q_lim <= 0) { skip_run(); continue; }
    q0 = max(0, q_lim - kBlockQ);
  } else {
    q_lim = qlen;
    q0 = (PACK ? 0 : q_tiles - 1 - qtile) * kBlockQ;
    if (q0 >= qlen) { skip_run(); continue; }
  }
  // keys: same rows as the queries (prefill), or their own length / base / page table (KV cache).  Causal masking is
  // bottom-right aligned when the key sequence is longer: query i sees keys <= i + (klen - qlen)  (flash-attn >= 2.1).
  const int len = pg.seqlens_k ? pg.seqlens_k[seq] : qlen;
  const int shift = len - qlen;
  const int ks = pg.cu_k ? pg.cu_k[seq] : qs;
  const int vq = PACK ? j / rep : q0 + qw * 32 + j;           // this lane's query
  const int my_q = min(vq, q_lim - 1);                        // clamped: padding lanes replay the tile's last query
  const bool q_valid = vq < q_lim;
  const int wave_qmax = PACK ? qlen - 1 : min(q0 + qw * 32 + 31, q_lim - 1);
  const int wave_qmin = PACK ? 0 : q0 + qw * 32;

  // Q^T fragments (B operand of S^T = K Q^T): lane (j, hh) holds q[my_q][16*step + 8*hh .. +7]
  u32x4 qf[kSteps];
  {
    const uint16_t* qp = q + (int64_t)qs * hq * kHeadDim + my_q * q_rs + head * q_hs + 8 * hh;
#pragma unroll
    for (int st = 0; st < kSteps; ++st) qf[st] = *reinterpret_cast<const u32x4*>(qp + 16 * st);
  }
  f32x16 o[kDTiles];
#pragma unroll
  for (int dt = 0; dt < kDTiles; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const int kv_end = causal ? max(0, min(len, (PACK ? qlen : (TAIL ? q_lim : q0 + kBlockQ)) + shift)) : len;
  const int n_tiles = (kv_end + kTileK - 1) / kTileK;
  if (n_tiles <= 0) {     // no visible key for this whole q-tile (key sequence shorter than the query offset): zeros
    if (q_valid) {
      uint16_t* op = out + ((int64_t)(qs + my_q) * hq + head) * kHeadDim;
#pragma unroll
      for (int c = 0; c < kParts; ++c)
        if ((c & 1) == hh) *reinterpret_cast<u32x4*>(op + 8 * c) = u32x4{0u, 0u, 0u, 0u};
      if (lse && hh == 0) lse[(int64_t)(qs + my_q) * hq + head] = -INFINITY;
    }
    continue;
  }

  const int64_t kv_base = (int64_t)ks * hkv * kHeadDim + kvh * kv_hs;            // contiguous keys: row r adds r * kv_rs
  const int64_t v_base = (int64_t)ks * hkv * kHeadDim + kvh * v_hs;
  // Fast path (contiguous keys, full tile): the address of a slice is a wave-uniform 64-bit row base (scalar ALU) plus a
  // per-lane 32-bit byte offset that is computed ONCE per item -- tools/trace_attn.py measured 0.9-1.3 us per tile spent
  // ISSUING the 16 direct-to-LDS loads of a 2-wave group when every load carried its own 64-bit multiplies and the branches
  // of the paged / tail form.
  constexpr int NP = GW >= 4 ? 1 : 4 / GW;                     // distinct swizzle keys (i & 3) among a wave's slices
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  uint32_t lane_off_k[NP], lane_off_v[NP];
#pragma unroll
  for (int pp = 0; pp < NP; ++pp) {
    const int part8 = src_part8(wave + GW * pp);
    lane_off_k[pp] = (uint32_t)(((int64_t)(lane >> 4) * kv_rs + part8) * 2);
    lane_off_v[pp] = (uint32_t)(((int64_t)(lane >> 4) * v_rs + part8) * 2);
  }
  auto load_tile = [&](int t, int buf) {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const bool tail = __builtin_amdgcn_readfirstlane((int)(t * kTileK + kTileK > len)) != 0;
    if (!pg.block_table && !tail) {
      // one 64-bit multiply per tensor and tile (the tile's first row of this wave), then a constant stride between slices
      const char* kp = reinterpret_cast<const char*>(k + kv_base + (int64_t)(t * kTileK + 4 * wave_u) * kv_rs);
      const char* vp = reinterpret_cast<const char*>(v + v_base + (int64_t)(t * kTileK + 4 * wave_u) * v_rs);
      const int64_t k_step = 2 * 4 * GW * kv_rs, v_step = 2 * 4 * GW * v_rs;          // bytes from slice i to slice i + GW
#pragma unroll
      for (int u = 0; u < kLoadsPerWave; ++u) {
        const int i = wave_u + GW * u;
        __builtin_amdgcn_global_load_lds((gptr_t)(kp + lane_off_k[u % NP]), (lptr_t)(k_sm + buf * kBuf + i * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(vp + lane_off_v[u % NP]), (lptr_t)(v_sm + buf * kBuf + i * 1024), 16, 0, 0);
        kp += k_step;
        vp += v_step;
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < kLoadsPerWave; ++u) {
      const int i = wave + GW * u;
      const int key = 4 * i + (lane >> 4);
      const int kpos = min(t * kTileK + key, len - 1);
      int64_t row;
      if (pg.block_table) {
        const int page = kpos / pg.page_size;
        row = (int64_t)pg.block_table[(int64_t)seq * pg.max_pages + page] * pg.page_size + (kpos - page * pg.page_size);
      } else {
        row = kpos;
      }
      const int64_t off = (pg.block_table ? (row * hkv + kvh) * kHeadDim : kv_base + row * kv_rs) + src_part8(i);
      const int64_t off_v = (pg.block_table ? (row * hkv + kvh) * kHeadDim : v_base + row * v_rs) + src_part8(i);
      __builtin_amdgcn_global_load_lds((gptr_t)(k + off), (lptr_t)(k_sm + buf * kBuf + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(v + off_v), (lptr_t)(v_sm + buf * kBuf + i * 1024), 16, 0, 0);
    }
  };
  // group g walks tiles g, g + KVS, ...; every wave executes the same number of rounds (and barriers)
  const int n_rounds = (n_tiles + KVS - 1) / KVS;
  if (grp < n_tiles) load_tile(grp, 0);
  __syncthreads();
  VSEL_STAMP(0, 1);

  [[maybe_unused]] bool trace_on = false;
  // one 64-key tile from LDS buffer CUR (compile-time, so that every LDS address is a per-lane base + an immediate)
  auto tile_body = [&](auto cur_c, int t) {
    constexpr int CUR = decltype(cur_c)::value;
    // a wave whose 32 query slots are all padding (short q-tiles: decode, ragged tails) only helps with the loads
    // (KH = 2: the wave's own 32-key block must hold a visible key)
    const int kblk0 = t * kTileK + 32 * kh;
    const bool wave_active = (wave_qmin < q_lim) && (!causal || (kblk0 <= wave_qmax + shift)) && (KH == 1 || kblk0 < len);
    if (wave_active) {
      const char* kt = smem + CUR * kBuf;
      const char* vt = smem + (2 + CUR) * kBuf;
      // ---- S^T = K Q^T ----------------------------------------------------------------------------------------
      f32x16 s[NKB];                    // s[b]: key block b (KH = 1) / the wave's own key block kh (KH = 2)
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
      if constexpr (USE_TR && kSteps == 8 && !PACK) {
        // per 32-key block: its 8 K fragments in ONE batch of ds_read_b128, then its 8 MFMAs (one LDS round trip per block instead
        // of one per MFMA; a second batch in flight would need 32 more registers and spills; batches of 4 with two in flight --
        // same registers, reads ahead of the MFMAs -- measured SLOWER, 880 vs 973 TFLOP/s at 16 x 4096)
        auto s_block = [&](auto kb_c) {
          constexpr int KB = decltype(kb_c)::value;
          constexpr int SI = KH == 1 ? KB : 0;
          if constexpr (NW >= 8) {
            u32x4 ka[8];
#pragma unroll
            for (int st = 0; st < 8; ++st) ka[st] = lds_read_b128_asm<CUR * kBuf + KB * 32 * kRowBytes>(row_addr_u[st]);
            lds_wait8<0>(ka);
#pragma unroll
            for (int st = 0; st < 8; ++st)
              s[SI] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(to_bf16x8(ka[st]), to_bf16x8(qf[st]), s[SI], 0, 0, 0);
          } else {
            // 4-wave workgroups stage twice the tile rows per wave (more address state): batches of 4 keep the kernel spill-free
#pragma unroll
            for (int h4 = 0; h4 < 2; ++h4) {
              u32x4 ka[4];
#pragma unroll
              for (int st = 0; st < 4; ++st) ka[st] = lds_read_b128_asm<CUR * kBuf + KB * 32 * kRowBytes>(row_addr_u[4 * h4 + st]);
              lds_wait4<0>(ka);
#pragma unroll
              for (int st = 0; st < 4; ++st)
                s[SI] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(to_bf16x8(ka[st]), to_bf16x8(qf[4 * h4 + st]), s[SI], 0, 0, 0);
            }
          }
        };
        s_block(std::integral_constant<int, 0>{});          // (KH = 2: the per-lane bases already point at the wave's own key block)
        if constexpr (KH == 1) s_block(std::integral_constant<int, 1>{});
      } else {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int st = 0; st < kSteps; ++st) {
            const u32x4 a = *reinterpret_cast<const u32x4*>(kt + row_addr[st] + kb * 32 * kRowBytes);
            s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(to_bf16x8(a), to_bf16x8(qf[st]), s[kb], 0, 0, 0);
          }
      }
      VSEL_FWD_STAMP(2);
      // the first group of V fragments does not depend on P: put its transpose reads in flight now, under the softmax
      u32x2 vr0[8], vr1[8];
      if constexpr (USE_TR && kDTiles == 4 && NW >= 8 && !PACK) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          vr0[2 * dt] = lds_read_tr16_b64_asm<(2 + CUR) * kBuf>(tr_addr_u[dt][0]);
          vr0[2 * dt + 1] = lds_read_tr16_b64_asm<(2 + CUR) * kBuf>(tr_addr_u[dt][1]);
        }
      }
      // ---- mask + online softmax (exp2 domain; the softmax scale is folded into one FMA per element) ------------
      // wave-uniform (made explicit with readfirstlane so the compiler emits ONE scalar branch, not an exec-mask dance per
      // element): only tiles that straddle the causal diagonal or the end of the key sequence pay for the mask
      const bool need_mask = __builtin_amdgcn_readfirstlane(
          (int)((t * kTileK + kTileK > len) || (causal && (t * kTileK + kTileK - 1 > wave_qmin + shift)))) != 0;
      float mx = -INFINITY;
      if (need_mask) {
        const int kmax = causal ? min(len - 1, my_q + shift) : len - 1;     // last visible key of this lane's query
        const int kbase0 = t * kTileK + 8 * hh + (KH == 1 ? 0 : 32 * kh);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kbase0 + 32 * kb + 16 * (r >> 3) + (r & 7);
            const float val = key <= kmax ? s[kb][r] : -INFINITY;
            s[kb][r] = val;
            mx = fmaxf(mx, val);
          }
      } else {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      // Lazy reference exponent: m_run is the exponent the row's (l, O) are expressed in, not necessarily the running maximum.
      // It moves only when the tile's maximum exceeds it by more than kLazyTau (log2 units), so p = exp2(s - m_run) <= 2^kLazyTau:
      // harmless in fp32 / bf16, and the max / rescale of O leaves the common path (with the exact running maximum some lane of
      // the wave moved in most of the first ~2000 keys of a row).  The decision is per lane (per query), so a row's arithmetic does
      // not depend on which other rows share its wave (PACK / NW forms stay bit-identical to each other).
      const float m_cand = fmaxf(m_run, mx * scale_log2e);     // scale > 0: max commutes with the scaling
      const bool moves = m_cand > m_run + kLazyTau;
      if (__any(moves)) {                                      // wave-uniform branch
        const float m_new = moves ? m_cand : m_run;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // exactly 1 for the lanes that stay
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < kDTiles; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      }
      float psum = 0.f;
      bf16x8_t pf[NKB][2];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(fmaf(s[kb][r], scale_log2e, -m_run));
          psum += p;
          pf[kb][r >> 3][r & 7] = (__bf16)p;
        }
      l_run += psum;
      VSEL_FWD_STAMP(3);
      // ---- O^T += V^T P^T -------------------------------------------------------------------------------------
      if constexpr (USE_TR && kDTiles == 4 && !PACK) {
        // four (32-key block, 16-key half) groups; a group = 8 transpose reads (4 d-tiles x lo / hi) feeding 4 MFMAs; the next
        // group's reads are in flight behind the current group's MFMAs
        auto issue = [&](auto g_c, u32x2 (&dst)[8]) {
          constexpr int G = decltype(g_c)::value;
          constexpr int OFF = (2 + CUR) * kBuf + (32 * (G >> 1) + 16 * (G & 1)) * kRowBytes;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            dst[2 * dt] = lds_read_tr16_b64_asm<OFF>(tr_addr_u[dt][0]);
            dst[2 * dt + 1] = lds_read_tr16_b64_asm<OFF>(tr_addr_u[dt][1]);
          }
        };
        auto pv = [&](auto g_c, u32x2 (&src)[8]) {
          constexpr int G = decltype(g_c)::value;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const u32x4 w = {src[2 * dt][0], src[2 * dt][1], src[2 * dt + 1][0], src[2 * dt + 1][1]};
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(to_bf16x8(w), pf[KH == 1 ? (G >> 1) : 0][G & 1], o[dt], 0, 0, 0);
          }
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        if constexpr (KH == 2) {
          // the wave's own key block = two 16-key groups (the first was issued before the softmax)
          issue(I1{}, vr1);
          lds_wait8<8>(vr0);
          pv(I0{}, vr0);
          lds_wait8<0>(vr1);
          pv(I1{}, vr1);
        } else if constexpr (NW >= 8) {
          issue(I1{}, vr1);                    // (group 0 was issued before the softmax)
          lds_wait8<8>(vr0);
          pv(I0{}, vr0);
          issue(I2{}, vr0);
          lds_wait8<8>(vr1);
          pv(I1{}, vr1);
          issue(I3{}, vr1);
          lds_wait8<8>(vr0);
          pv(I2{}, vr0);
          lds_wait8<0>(vr1);
          pv(I3{}, vr1);
        } else {
          issue(I0{}, vr0); lds_wait8<0>(vr0); pv(I0{}, vr0);
          issue(I1{}, vr0); lds_wait8<0>(vr0); pv(I1{}, vr0);
          issue(I2{}, vr0); lds_wait8<0>(vr0); pv(I2{}, vr0);
          issue(I3{}, vr0); lds_wait8<0>(vr0); pv(I3{}, vr0);
        }
      } else {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
          const int off = (32 * kb + 16 * mm) * kRowBytes;
#pragma unroll
          for (int dt = 0; dt < kDTiles; ++dt) {
            bf16x8_t vf;
            if constexpr (USE_TR) {
              // 16-lane group: lane p supplies the address of row (p >> 2), 4-column chunk (p & 3) of a [4 keys x 16 d]
              // block and receives column p (4 keys) -- hardware transpose read
              typedef __attribute__((address_space(3))) bf16x4_t* lds_p;
              const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(vt + tr_addr[dt][0] + off));
              const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(vt + tr_addr[dt][1] + off));
              vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            } else {
              const int dcol = 32 * dt + j;
#pragma unroll
              for (int e = 0; e < 8; ++e)
                vf[e] = *reinterpret_cast<const __bf16*>(vt + chunk_off(32 * kb + 16 * mm + 8 * hh + e, dcol >> 3) + (dcol & 7) * 2);
            }
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb][mm], o[dt], 0, 0, 0);
          }
        }
      }
    }
  };
  if constexpr (KVS == 1) {
    for (int t = 0; t < n_tiles; t += 2) {
#ifdef VSEL_TRACE
      trace_on = blockIdx.x == 0 && round == 0 && t == ((n_tiles / 2) & ~1);
#endif
      VSEL_FWD_STAMP(0);
      if (t + 1 < n_tiles) load_tile(t + 1, 1);
      VSEL_FWD_STAMP(1);
      tile_body(std::integral_constant<int, 0>{}, t);
      VSEL_FWD_STAMP(4);
      __syncthreads();                       // also drains this wave's global_load_lds queue (vmcnt(0)) before the release
      VSEL_FWD_STAMP(5);
#ifdef VSEL_TRACE
      trace_on = false;
#endif
      if (t + 1 >= n_tiles) break;
      if (t + 2 < n_tiles) load_tile(t + 2, 0);
      tile_body(std::integral_constant<int, 1>{}, t + 1);
      __syncthreads();
    }
  } else {
    for (int r = 0; r < n_rounds; r += 2) {
      const int t0 = KVS * r + grp, t1 = t0 + KVS, t2 = t1 + KVS;
      if (t1 < n_tiles) load_tile(t1, 1);
      VSEL_STAMP(3, min(r, 7));
      if (t0 < n_tiles) tile_body(std::integral_constant<int, 0>{}, t0);
      VSEL_STAMP(2, min(r, 7));
      __syncthreads();
      VSEL_STAMP(1, min(r, 7));
      if (r + 1 >= n_rounds) break;
      if (t2 < n_tiles) load_tile(t2, 0);
      VSEL_STAMP(3, min(r + 1, 7));
      if (t1 < n_tiles) tile_body(std::integral_constant<int, 1>{}, t1);
      VSEL_STAMP(2, min(r + 1, 7));
      __syncthreads();
      VSEL_STAMP(1, min(r + 1, 7));
    }
  }
  VSEL_STAMP(0, 2);

  constexpr int P = KVS * KH;              // partial online-softmax states per 32-query slice
  if constexpr (KVS == 2) {
    // merge the P partial states of a slice (one per KV stream and key half), all P waves at once: partial p OWNS the d-tiles
    // dt with dt % P == p.  Every wave publishes (m, l) and the O rows of the d-tiles it does not own through LDS (the tile
    // buffers are free now), then rescales its own d-tiles to the common maximum, adds the others' in index order and stores
    // those columns.  (One wave merging everything serially and storing whole rows cost 3.1 us of the 15 at L' = 524.)
    static_assert(kDTiles % P == 0, "d-tiles are dealt out evenly to the partials");
    float* ex = reinterpret_cast<float*>(smem);
    constexpr int kPub = 2 + 16 * (kDTiles - kDTiles / P);      // floats a wave publishes per lane
    static_assert((size_t)P * QW * kPub * 64 * sizeof(float) <= (size_t)KVS * kLds, "merge area exceeds the tile buffers");
    auto pub_pos = [&](int p, int dt) {      // index of d-tile dt (not owned by p) among p's published d-tiles
      return (dt / P) * (P - 1) + (dt % P) - ((dt % P) > p ? 1 : 0);
    };
    __syncthreads();
    {
      float* mine = ex + ((size_t)(pidx * QW + qw) * kPub) * 64 + lane;
      mine[0] = m_run;
      mine[64] = l_run;
#pragma unroll
      for (int dt = 0; dt < kDTiles; ++dt) {
        if (dt % P == pidx) continue;        // uniform
        const int pos = pub_pos(pidx, dt);
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) mine[(2 + pos * 16 + rr) * 64] = o[dt][rr];
      }
    }
    __syncthreads();
    float a[P];
    float m_max = m_run;
#pragma unroll
    for (int pp = 0; pp < P; ++pp) {
      a[pp] = ex[((size_t)(pp * QW + qw) * kPub) * 64 + lane];                  // m of partial pp
      m_max = fmaxf(m_max, a[pp]);
    }
    float l_sum = 0.f;
#pragma unroll
    for (int pp = 0; pp < P; ++pp) {
      a[pp] = __builtin_amdgcn_exp2f(a[pp] - m_max);
      l_sum += ex[((size_t)(pp * QW + qw) * kPub + 1) * 64 + lane] * a[pp];
    }
    m_run = m_max;
    l_run = l_sum;
#pragma unroll
    for (int dt = 0; dt < kDTiles; ++dt) {
      if (dt % P != pidx) continue;          // uniform
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        float acc = 0.f;
#pragma unroll
        for (int pp = 0; pp < P; ++pp) {
          const float v = (pp == pidx) ? o[dt][rr]
                                       : ex[((size_t)(pp * QW + qw) * kPub + 2 + pub_pos(pp, dt) * 16 + rr) * 64 + lane];
          acc += v * a[pp];
        }
        o[dt][rr] = acc;
      }
    }
  }

  VSEL_STAMP(0, 3);
  // ---- epilogue: O^T[d][query] / l, 4 consecutive d per store ------------------------------------------------------
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;      // a row that sees no key (klen < qlen) outputs zeros
  if (q_valid) {
    // log-sum-exp of the scaled scores (natural log), saved for the backward pass; -inf for a row without a visible key
    if (lse && hh == 0 && pidx == 0)
      lse[(int64_t)(qs + my_q) * hq + head] = l_tot > 0.f ? (m_run + log2f(l_tot)) * 0.6931471805599453f : -INFINITY;
    uint16_t* op = out + ((int64_t)(qs + my_q) * hq + head) * kHeadDim;
#pragma unroll
    for (int dt = 0; dt < kDTiles; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        if (P > 1 && dt % P != pidx) continue;        // (two-stream form: every partial stores the d-tiles it owns)
        const int d0 = 32 * dt + 8 * g4 + 4 * hh;
        if (d0 >= kHeadDim) continue;                 // head_dim 80: the last d-tile is half padding
        // (v_cvt_pk_bf16_f32 = round to nearest even, the software rounding bit for bit on finite values: 2 instructions per four values
        // where f32_to_bf16_bits + the packing were ~22)
        uint2 pk;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk.x) : "v"(o[dt][4 * g4] * inv), "v"(o[dt][4 * g4 + 1] * inv));
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk.y) : "v"(o[dt][4 * g4 + 2] * inv), "v"(o[dt][4 * g4 + 3] * inv));
        *reinterpret_cast<uint2*>(op + d0) = pk;
      }
  }
  VSEL_STAMP_DRAIN(0, 4);
  }  // persistent item loop
}

}  // namespace vsel

using namespace vsel;

#ifdef VSEL_TRACE
extern "C" int vsel_debug_read_fwd_tile_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fwd_tile_trace), sizeof(g_fwd_tile_trace)) == hipSuccess ? VSEL_OK : VSEL_ERR_HIP;
}
// copies the stamps of the last forward launch: out[kTraceKernels][kTraceBlocks][kTraceSlots] (tools/trace_attn.py)
extern "C" int vsel_debug_read_attn_trace(unsigned long long* out, int clear) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(g_trace)) != hipSuccess) return VSEL_ERR_HIP;
  if (clear) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_trace)) != hipSuccess || hipMemset(p, 0, sizeof(g_trace)) != hipSuccess) return VSEL_ERR_HIP;
  }
  return VSEL_OK;
}
#endif

// form selection knobs (include/vsel_debug.h; defaults in common.hip):
//   ATTN_WAVES 0 = choose by grid size, 4 / 8 = force the workgroup size
//   ATTN_PACK  0 = never, 1 = whenever qlen * rep <= 32, 2 (default) = against a cache when the per-head grid exceeds one round
//   ATTN_SPLIT 0 = never, 1 = whenever the 4-wave grid has <= 256 items, 2 (default) = ... and the sequences are not tiny
//   ATTN_SPLIT_Q64 64-query workgroups for the two-stream form when they fit one per CU

static int attn_launch(hipStream_t st, const void* q, const void* k, const void* v, const int32_t* cu_q, int64_t n_seq,
                       int64_t max_seqlen_q, int64_t hq, int64_t hkv, int64_t d, float scale, int causal, void* out,
                       const PagedKV& pg, float* lse = nullptr, int64_t total = 0) {
  const bool g_attn_use_tr = knob(VSEL_KNOB_ATTN_USE_TR) != 0, g_attn_split_q64 = knob(VSEL_KNOB_ATTN_SPLIT_Q64) != 0;
  const int g_attn_nw = knob(VSEL_KNOB_ATTN_WAVES), g_attn_pack = knob(VSEL_KNOB_ATTN_PACK), g_attn_split = knob(VSEL_KNOB_ATTN_SPLIT);
  // 8-wave workgroups (256 queries) when there is enough work to fill the chip with them, else 4-wave (128 queries)
  const int64_t items8 = cdiv(max_seqlen_q, 256) * hq * n_seq;
  // measured on MI355X (tools/exp_attn_nw.py): 8 waves +12-16 % at L >= 4096, +4 % at 16 x 2368, -20 % at L = 524
  // ... and for one or two long sequences (same-box scan, 4 vs 8 waves: 1 x 4096 151 vs 133 us, 1 x 8192 498 vs 463, 2 x 4096 266 vs 253;
  // at 2368-3000 tokens per sequence the two forms are within 2 %)
  const bool big = g_attn_nw == 8 || (g_attn_nw == 0 && ((max_seqlen_q >= 2048 && items8 >= 1024) || (max_seqlen_q >= 4096 && items8 >= 400)));
  // decode / short chunks against a cache with d = 128: one wave per (kv head, sequence) serving the whole GQA group
  // (measured, tools/bench_decode.py, 28 / 4 heads, Lk = 524: 2.1x at 64 sequences, 2.75x at 256; but a lone wave stages whole
  // tiles by itself, so below one round of per-head workgroups -- n_seq * hq <= 512 -- the per-head form is faster: 24 vs 41 us)
  const bool pack = g_attn_pack != 0 && d == 128 && g_attn_nw == 0 && max_seqlen_q * (hq / hkv) <= 32 &&
                    (g_attn_pack == 1 || (pg.seqlens_k != nullptr && n_seq * hq > 512));
  // one short sequence (or a few): the grid does not fill the chip and the cost is the dependent chain of KV tiles -> two
  // KV streams per workgroup (tools/bench_attn.py: L' = 524 19.5 -> see profiles/r01_attention.txt)
  const int64_t items4 = cdiv(max_seqlen_q, 128) * hq * n_seq;
  // (not for the LSE-saving training forward: the split changes the summation order, and training keeps "a sequence's
  // results do not depend on what it is packed with" bit for bit)
  const bool split2 = g_attn_split != 0 && d == 128 && !pack && !big && g_attn_nw == 0 && g_attn_use_tr && items4 <= 256 &&
                      (g_attn_split == 1 || (lse == nullptr && (pg.seqlens_k != nullptr || max_seqlen_q >= 256)));
  // ... and 64-query workgroups whose wave pairs split every tile's keys (KH = 2) while those still fit one per CU
  const bool split2_q64 = split2 && g_attn_split_q64 && cdiv(max_seqlen_q, 64) * hq * n_seq <= 256;
  // 64 rows per wave, one wave per SIMD, hand-scheduled tile loop (attn_fwd64.hip): head_dim 128 from 2048 tokens in the longest
  // sequence, whatever the batch (same-process A/B, tools/exp_fwd64_shapes.py, profiles/r04_fwd64_shapes.txt: 1 x 2368 59.8 vs 68.0 us,
  // 1 x 4096 112 vs 125, 8 x 2048 +15 %, 16 x 4096 +11 %, 2 x 8192 +16 %; 16 x 1100 -2 %, 32 x 524 -20 %: those keep the 4-wave form)
  const int g_rows64 = knob(VSEL_KNOB_ATTN_ROWS64);
  // Short sequences (packed batches of compressed prompts): ONE workgroup per (query tile, kv head) serves the whole q-head group, so a
  // K / V tile is loaded once per group.  Prefill over the queries' own contiguous keys only.  Two forms, bit-identical:
  //   * attn_fwd_gqa.hip: 8 waves = the group's heads on a 32-query tile, items pipelined into each other (next item's rows in flight under
  //     the last tile, queue drawn ahead) -- best when a workgroup runs many items (>= 3 rounds of them);
  //   * attn_fwd_gqa64.hip: attn_fwd64's generated loop with two heads per wave (~8 instead of ~18 instructions per MFMA, but every item
  //     pays its own prologue) -- best for few items per workgroup and for groups of <= 4 heads (64-query items).
  // Same-process A/B (profiles/r05_gqa_ab.txt, second table; us per-head / 8-wave dealt / generated dealt, 28 / 4 heads): 4 x 524 25.8 / 23.9 /
  // 21.5, 5 x 524 31.7 / 25.3 / 24.6, 6 x 524 35.3 / 25.2 / 27.1, 8 x 524 43.0 / 32.6 / 36.8, 32 x 524 146 / 110 / 128, 32 x 294 70.5 / 51.3 / 64.7,
  // 16 x 1100 219 / 188.5 / 186, 32 x 1216 486 / 448 / 443; ragged (queue behind two dealt rounds): 8 prompts 52.1 / 51.2 / 51.2, 64 prompts 325 /
  // 276 / 288; LLaVA-OV 32 / 8 heads 8 x 1230 148 / 122 / 115, 32 x 1230 589 / 521 / 479; 3B 16 / 2 heads 32 x 524 88 / 59 / 68.  Dealing the
  // list (mirrored rounds) instead of drawing it an item or two ahead is what moved the few-round grids: 8 x 524 was 44.7 / 44.3 drawn.
  // One sequence (68 items) stays with the two-stream per-head form (13.5 vs 19 us); from 2048 tokens attn_fwd64 (256-query items).
  const int g_gqa = knob(VSEL_KNOB_ATTN_GQA);
  const int64_t rep_ = hq / hkv;
  if (d == 128 && g_attn_use_tr && !pg.block_table && !pg.seqlens_k && !pg.cu_k && g_attn_nw == 0 && rep_ >= 2 && rep_ <= 8 &&
      max_seqlen_q <= 16384 && g_rows64 != 1) {
    // items of the 8-wave form that hold work (ragged batches: from the token count when the entry point knows it -- the item LIST is
    // sized by the longest sequence, but a shorter sequence's items at the levels it does not reach are skipped in runs)
    const int64_t bq8 = 32 * (8 / rep_);
    const int64_t gqa_items8 = (total > 0 ? std::min(cdiv(total, bq8) + n_seq, cdiv(max_seqlen_q, bq8) * n_seq) : cdiv(max_seqlen_q, bq8) * n_seq) * hkv;
    const int64_t bq64 = 32 * (4 / ((rep_ + 1) / 2));
    const int64_t gqa_items64 = (total > 0 ? std::min(cdiv(total, bq64) + n_seq, cdiv(max_seqlen_q, bq64) * n_seq) : cdiv(max_seqlen_q, bq64) * n_seq) * hkv;
    // which form (-1: none).  The generated loop for groups of <= 4 heads (64-query items), for long-ish sequences (>= 1536 tokens: the tile
    // loop dominates) and for grids of one or two rounds of items; the 8-wave form from 1.6 rounds of its items when they are dealt
    // (uniform batches), from three rounds when they are drawn (ragged ones).
    const bool uniform = total > 0 && total == n_seq * max_seqlen_q;
    int form = -1;
    if (g_gqa == 1) form = knob(VSEL_KNOB_ATTN_GQA_FORM) == 1 ? 1 : (knob(VSEL_KNOB_ATTN_GQA_FORM) == 0 ? 0 : (rep_ <= 4 ? 1 : 0));
    else if (g_gqa < 0 && max_seqlen_q < 2048 && !split2) {
      if ((rep_ <= 4 || max_seqlen_q >= (uniform ? 1024 : 1536)) && gqa_items64 >= 128) form = 1;     // (dealt: 16 x 1100 182 vs 189 us, 32 x 1216 430 vs 448)
      else if (gqa_items8 >= (uniform ? 400 : 768)) form = 0;       // (dealt: from 1.6 rounds of its items -- 6 x 524 25.5 us against 26.8 / 36.2)
      else if (gqa_items64 >= 128 && 20 * gqa_items64 <= 36 * 256) form = 1;
      if (form >= 0 && knob(VSEL_KNOB_ATTN_GQA_FORM) >= 0) form = knob(VSEL_KNOB_ATTN_GQA_FORM);
    }
    if (form >= 0) {
      // uniform batches: the item list dealt out with mirrored rounds (the queue's draws run one or two items ahead, i.e. in arrival order
      // for the first rounds, and every workgroup would keep its level's rank); ragged ones keep the queue, whose counter jumps over empty
      // runs -- with the second round dealt as well
      const int g_static = knob(VSEL_KNOB_ATTN_STATIC);
      const bool deal = g_static == 1 || (g_static < 0 && uniform);
      return (form == 1 ? attn::attn_fwd_gqa64_launch : attn::attn_fwd_gqa_launch)(st, q, k, v, cu_q, n_seq, max_seqlen_q, hq, hkv, scale, causal,
                                                                                   out, pg, lse, deal);
    }
  }
  if (d == 128 && g_attn_use_tr && !pack && !pg.block_table && g_attn_nw == 0 &&
      (g_rows64 == 1 || (g_rows64 < 0 && max_seqlen_q >= 2048 && !split2)))
    return attn::attn_fwd64_launch(st, q, k, v, cu_q, n_seq, max_seqlen_q, hq, hkv, scale, causal, out, pg, lse);
  const int block_q = big ? 256 : (split2_q64 ? 64 : 128);
  const int q_tiles = (int)cdiv(max_seqlen_q, block_q);
  const int64_t n_items = pack ? hkv * n_seq : (int64_t)q_tiles * hq * n_seq;
  if (n_items >= (1ll << 31)) return fail(VSEL_ERR_UNSUPPORTED, "too many attention work items");
  const int64_t slots = (big || split2) ? 256 : 512;               // resident workgroups (64 KiB LDS each; 128 KiB with two streams)
  int slot = -1, taken = -1;                                       // -1: direct mapping, one item per workgroup; a counter slot per queued launch
  // Few rounds of items (slots < n_items <= 2.35 slots, 4-wave form): the work queue's atomic round trip and hand-over barriers in front of
  // every item cost more than its balancing wins -- the heaviest-first list is dealt out statically instead, alternate rounds mirrored.
  // tools/exp_attn_static.py, profiles/r04_attn_static.txt (queue -> static, us): 4 x 524 30.6 -> 23.6, 6 x 524 41.6 -> 35.6, 8 x 524
  // 47.0 -> 42.0, 3 x 1100 53.1 -> 43.6, 4 x 1100 64.0 -> 57.4, 2 x 2000 85.5 -> 77.1, ragged batches of 3 - 6 compressed prompts +0 ... +18 %;
  // beyond that the queue wins (10 x 524 -3 %, 32 x 524 -7 %, 64 ragged prompts -14 %).  Also measured there and not kept: the next item's
  // Q rows prefetched into L2 by one load per lane (-1 ... -5 %: the load sits in front of the next tile's in the in-order return queue).
  if (!pack && !split2 && attn_static_deal(n_items, slots, !big && d == 128)) {
    slot = -2;                                                     // static deal (knob attn_static)
  } else if (n_items > slots) {
    if (int rc = queue_slot_acquire(kSlotFwd, st, &taken)) return rc;
    slot = taken;
    int* counters = nullptr;
    VSEL_HIP_CHECK(hipGetSymbolAddress((void**)&counters, HIP_SYMBOL(g_attn_work_counter)));
    VSEL_HIP_CHECK(hipMemsetAsync(counters + 8 * slot, 0, 8 * sizeof(int), st));
    // forward: +1.4 ... +3 % from 2048 tokens x 16 pairs (16 x 4096 982 -> 1011 TFLOP/s, 4 x 8192 1070 -> 1099, 16 x 2368 810 -> 822,
    // 4 x 2368 746 -> 760); one sequence (4 pairs) -12 %, 524-token sequences -40 %: those keep the single queue
    if (!pack && attn_use_xcd_queues(max_seqlen_q, n_seq * hkv, 2048, 16)) slot |= 0x100;
    else if (!pack && n_seq > 1 && knob(VSEL_KNOB_ATTN_SKIP_EMPTY) != 0) slot |= 0x200;
  }
  const dim3 grid((unsigned)std::min<int64_t>(n_items, slots));
  const float sl2 = scale * 1.4426950408889634f;
  // query tiles aligned to the end of each sequence when the 4-wave grid is throughput-bound (more than two rounds of items): +4-5 %
  // on packed batches of short / medium sequences (8 x 524, 32 x 524, 4 x 2368); -3 to -8 % when one sequence's longest chain is the
  // run time; the 8-wave instantiation measured -3 % at 16 x 4096 (where both tilings are the same tiles) and +2 % at 16 x 2368, so
  // it keeps the start-aligned tiling unless forced (same-box A/B, tools/ab_attn.sh)
  const int g_tail = knob(VSEL_KNOB_ATTN_TAIL_FIRST);
  const bool tail_first = causal && !pack && !split2 && d == 128 && g_attn_use_tr && (g_tail == 1 || (g_tail < 0 && !big && n_items > 2 * slots));
#define VSEL_ATTN_LAUNCH(TR, NWV, DV)                                                                                          \
  VSEL_LAUNCH((varlen_attn_fwd_kernel<TR, NWV, DV>), grid, dim3(64 * NWV), 0, st, (const uint16_t*)q, (const uint16_t*)k, \
                     (const uint16_t*)v, cu_q, (int)hq, (int)hkv, sl2, causal, (uint16_t*)out, q_tiles, (int)n_seq, slot, pg, lse)
  if (split2_q64) {
    VSEL_LAUNCH((varlen_attn_fwd_kernel<true, 8, 128, false, 2, 2>), grid, dim3(512), 0, st, (const uint16_t*)q,
                       (const uint16_t*)k, (const uint16_t*)v, cu_q, (int)hq, (int)hkv, sl2, causal, (uint16_t*)out, q_tiles,
                       (int)n_seq, slot, pg, lse);
  } else if (split2) {
    VSEL_LAUNCH((varlen_attn_fwd_kernel<true, 8, 128, false, 2>), grid, dim3(512), 0, st, (const uint16_t*)q,
                       (const uint16_t*)k, (const uint16_t*)v, cu_q, (int)hq, (int)hkv, sl2, causal, (uint16_t*)out, q_tiles,
                       (int)n_seq, slot, pg, lse);
  } else if (pack) {
    VSEL_LAUNCH((varlen_attn_fwd_kernel<true, 1, 128, true>), grid, dim3(64), 0, st, (const uint16_t*)q, (const uint16_t*)k,
                       (const uint16_t*)v, cu_q, (int)hq, (int)hkv, sl2, causal, (uint16_t*)out, q_tiles, (int)n_seq, slot, pg, lse);
  } else if (d == 128) {
    if (g_attn_use_tr && tail_first) {
#define VSEL_ATTN_LAUNCH_TAIL(NWV)                                                                                                  \
  VSEL_LAUNCH((varlen_attn_fwd_kernel<true, NWV, 128, false, 1, 1, true>), grid, dim3(64 * NWV), 0, st, (const uint16_t*)q, \
                     (const uint16_t*)k, (const uint16_t*)v, cu_q, (int)hq, (int)hkv, sl2, causal, (uint16_t*)out, q_tiles,        \
                     (int)n_seq, slot, pg, lse)
      if (big) VSEL_ATTN_LAUNCH_TAIL(8); else VSEL_ATTN_LAUNCH_TAIL(4);
#undef VSEL_ATTN_LAUNCH_TAIL
    } else if (g_attn_use_tr) {
      if (big) VSEL_ATTN_LAUNCH(true, 8, 128); else VSEL_ATTN_LAUNCH(true, 4, 128);
    } else {
      if (big) VSEL_ATTN_LAUNCH(false, 8, 128); else VSEL_ATTN_LAUNCH(false, 4, 128);
    }
  } else if (d == 80) {          // Qwen2.5-VL vision tower
    if (big) VSEL_ATTN_LAUNCH(true, 8, 80); else VSEL_ATTN_LAUNCH(true, 4, 80);
  } else {                       // d == 64: Rice ViT (LLaVA-OV-1.5)
    if (big) VSEL_ATTN_LAUNCH(true, 8, 64); else VSEL_ATTN_LAUNCH(true, 4, 64);
  }
#undef VSEL_ATTN_LAUNCH
  queue_slot_launched(kSlotFwd, taken, st);
  VSEL_AFTER_LAUNCH(st, "varlen_attn_fwd_kernel");
  return VSEL_OK;
}

static int attn_checks(const void* q, const void* k, const void* v, const int32_t* cu, const void* out, int64_t n_seq,
                       int64_t max_seqlen, int64_t hq, int64_t hkv, int64_t d) {
  if (!q || !k || !v || !cu || !out) return fail(VSEL_ERR_INVALID, "NULL pointer");
  if (d != 128 && d != 80 && d != 64) return fail(VSEL_ERR_UNSUPPORTED, "head_dim %lld not in {64, 80, 128}", (long long)d);
  if (n_seq < 1 || max_seqlen < 1 || hq < 1 || hkv < 1 || hq % hkv != 0 || n_seq > (1 << 24) || hq > 65535)
    return fail(VSEL_ERR_INVALID, "bad attention shape (n_seq=%lld max_seqlen=%lld hq=%lld hkv=%lld)", (long long)n_seq,
                (long long)max_seqlen, (long long)hq, (long long)hkv);
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) return fail(VSEL_ERR_INVALID, "q/k/v/out must be 16-byte aligned");
  return VSEL_OK;
}

extern "C" int vsel_varlen_attn_fwd(void* stream, const void* q, const void* k, const void* v, const int32_t* cu_seqlens,
                                    int64_t n_seq, int64_t max_seqlen, int64_t total, int64_t hq, int64_t hkv, int64_t d,
                                    float scale, int causal, void* out) {
  int rc = attn_checks(q, k, v, cu_seqlens, out, n_seq, max_seqlen, hq, hkv, d);
  if (rc) return rc;
  if (total < 1) return fail(VSEL_ERR_INVALID, "total must be >= 1");
  hipStream_t st = (hipStream_t)stream;
  VSEL_PROF_BEGIN(st);
  return attn_launch(st, q, k, v, cu_seqlens, n_seq, max_seqlen, hq, hkv, d, scale, causal, out, PagedKV{nullptr, nullptr, nullptr, 0, 1, 0, 0, 0, 0, 0, 0},
                     nullptr, total);
}

// ---- one long sequence (or a few of equal length): key-range parts with a caller workspace (attn_fwd64_parts.hip) -----------------------
// cap = key tiles per part: the launch's tile steps spread over the CUs, never below 8 (a part costs an item's prologue and epilogue)
static bool parts_plan_for(int64_t n_seq, int64_t max_seqlen, int64_t total, int64_t hq, int64_t hkv, int64_t d, int causal, attn::PartsPlan* plan) {
  const int g = knob(VSEL_KNOB_ATTN_KEY_PARTS);
  if (g == 0 || d != 128 || !causal || total != n_seq * max_seqlen || hq % hkv != 0) return false;
  const int64_t q_tiles = cdiv(max_seqlen, 256);
  const int64_t items = q_tiles * hq * n_seq;
  // by itself only for launches of at most half a round of items -- few q heads: small models, heads sharded over GPUs.  Measured (one
  // sequence, unsplit / parts us): 4 heads 2368 33.3 / 31.2, 8192 137.9 / 102.4; 8 heads 2368 43.5 / 35.4, 4096 72.2 / 68.8; but 12 heads
  // 2368 47.1 / 54.7, 28 heads 2368 60.8 / 79.9, 4096 112 / 139: the fp32 partials of 28 heads are 49 MB written and read again (merge
  // launch 17 us), more than the balance wins (profiles/EXPERIMENTS.md, round 5)
  if (g < 0 && (max_seqlen < 2048 || items > 128)) return false;
  int64_t steps = 0;                                                        // 64-key tile steps of the launch
  for (int64_t t = 0; t < q_tiles; ++t) steps += cdiv(std::min(max_seqlen, (t + 1) * 256), 64);
  steps *= hq * n_seq;
  const int cap = g > 1 ? g : (int)std::max<int64_t>(8, cdiv(steps * 3, 4 * 256));
  return attn::fwd64_parts_plan(n_seq, max_seqlen, hq, cap, plan);
}

extern "C" size_t vsel_varlen_attn_fwd_workspace_bytes(int64_t n_seq, int64_t max_seqlen, int64_t total, int64_t hq, int64_t hkv, int64_t d,
                                                       int causal) {
  attn::PartsPlan plan;
  if (n_seq < 1 || max_seqlen < 1 || hq < 1 || hkv < 1 || !parts_plan_for(n_seq, max_seqlen, total, hq, hkv, d, causal, &plan)) return 0;
  return attn::fwd64_parts_workspace_bytes(n_seq, max_seqlen, hq, plan.max_parts);
}

extern "C" int vsel_varlen_attn_fwd_ws(void* stream, const void* q, const void* k, const void* v, const int32_t* cu_seqlens, int64_t n_seq,
                                       int64_t max_seqlen, int64_t total, int64_t hq, int64_t hkv, int64_t d, float scale, int causal,
                                       void* out, float* lse, void* ws, size_t ws_bytes) {
  int rc = attn_checks(q, k, v, cu_seqlens, out, n_seq, max_seqlen, hq, hkv, d);
  if (rc) return rc;
  if (total < 1) return fail(VSEL_ERR_INVALID, "total must be >= 1");
  hipStream_t st = (hipStream_t)stream;
  VSEL_PROF_BEGIN(st);
  attn::PartsPlan plan;
  if (ws && parts_plan_for(n_seq, max_seqlen, total, hq, hkv, d, causal, &plan))
    return attn::attn_fwd64_parts_launch(st, q, k, v, n_seq, max_seqlen, hq, hkv, scale, out, lse, plan, ws, ws_bytes);
  return attn_launch(st, q, k, v, cu_seqlens, n_seq, max_seqlen, hq, hkv, d, scale, causal, out, PagedKV{nullptr, nullptr, nullptr, 0, 1, 0, 0, 0, 0, 0, 0},
                     lse, total);
}

extern "C" int vsel_varlen_attn_fwd_lse(void* stream, const void* q, const void* k, const void* v, const int32_t* cu_seqlens,
                                        int64_t n_seq, int64_t max_seqlen, int64_t total, int64_t hq, int64_t hkv, int64_t d,
                                        float scale, int causal, void* out, float* lse) {
  int rc = attn_checks(q, k, v, cu_seqlens, out, n_seq, max_seqlen, hq, hkv, d);
  if (rc) return rc;
  if (total < 1 || !lse) return fail(VSEL_ERR_INVALID, "total must be >= 1 and lse non-NULL");
  hipStream_t st = (hipStream_t)stream;
  VSEL_PROF_BEGIN(st);
  return attn_launch(st, q, k, v, cu_seqlens, n_seq, max_seqlen, hq, hkv, d, scale, causal, out,
                     PagedKV{nullptr, nullptr, nullptr, 0, 1, 0, 0, 0, 0, 0, 0}, lse, total);
}

extern "C" int vsel_varlen_attn_fwd_kv(void* stream, const void* q, const void* k, const void* v, const int32_t* cu_seqlens_q,
                                       const int32_t* cu_seqlens_k, const int32_t* seqlens_k, int64_t n_seq,
                                       int64_t max_seqlen_q, int64_t hq, int64_t hkv, int64_t d, float scale, int causal,
                                       void* out) {
  int rc = attn_checks(q, k, v, cu_seqlens_q, out, n_seq, max_seqlen_q, hq, hkv, d);
  if (rc) return rc;
  if (!cu_seqlens_k || !seqlens_k) return fail(VSEL_ERR_INVALID, "cu_seqlens_k / seqlens_k is NULL");
  hipStream_t st = (hipStream_t)stream;
  VSEL_PROF_BEGIN(st);
  return attn_launch(st, q, k, v, cu_seqlens_q, n_seq, max_seqlen_q, hq, hkv, d, scale, causal, out,
                     PagedKV{seqlens_k, nullptr, cu_seqlens_k, 0, 1, 0, 0, 0, 0, 0, 0});
}

extern "C" int vsel_varlen_attn_fwd_strided(void* stream, const void* q, const void* k, const void* v, const int32_t* cu_seqlens_q,
                                            const int32_t* cu_seqlens_k, const int32_t* seqlens_k, int64_t n_seq,
                                            int64_t max_seqlen_q, int64_t hq, int64_t hkv, int64_t d, int64_t q_row_stride,
                                            int64_t q_head_stride, int64_t k_row_stride, int64_t k_head_stride,
                                            int64_t v_row_stride, int64_t v_head_stride, float scale, int causal, void* out) {
  int rc = attn_checks(q, k, v, cu_seqlens_q, out, n_seq, max_seqlen_q, hq, hkv, d);
  if (rc) return rc;
  if ((cu_seqlens_k == nullptr) != (seqlens_k == nullptr)) return fail(VSEL_ERR_INVALID, "give cu_seqlens_k and seqlens_k together");
  if (q_row_stride < d || q_head_stride < d || k_row_stride < d || k_head_stride < d || v_row_stride < d || v_head_stride < d ||
      (q_row_stride | q_head_stride | k_row_stride | k_head_stride | v_row_stride | v_head_stride) % 8)
    return fail(VSEL_ERR_INVALID, "strides must be >= head_dim and multiples of 8 elements (16-byte rows)");
  hipStream_t st = (hipStream_t)stream;
  VSEL_PROF_BEGIN(st);
  PagedKV pg{seqlens_k, nullptr, cu_seqlens_k, 0, 1, q_row_stride, q_head_stride, k_row_stride, k_head_stride, v_row_stride, v_head_stride};
  return attn_launch(st, q, k, v, cu_seqlens_q, n_seq, max_seqlen_q, hq, hkv, d, scale, causal, out, pg);
}

extern "C" int vsel_paged_attn_fwd(void* stream, const void* q, const void* k_cache, const void* v_cache,
                                   const int32_t* cu_seqlens_q, const int32_t* seqlens_k, const int32_t* block_table,
                                   int64_t max_pages_per_seq, int64_t page_size, int64_t n_seq, int64_t max_seqlen_q, int64_t hq,
                                   int64_t hkv, int64_t d, float scale, int causal, void* out) {
  int rc = attn_checks(q, k_cache, v_cache, cu_seqlens_q, out, n_seq, max_seqlen_q, hq, hkv, d);
  if (rc) return rc;
  if (!seqlens_k || !block_table) return fail(VSEL_ERR_INVALID, "seqlens_k / block_table is NULL");
  if (page_size < 1 || page_size > (1 << 20) || max_pages_per_seq < 1 || max_pages_per_seq > (1ll << 31) - 1)
    return fail(VSEL_ERR_INVALID, "bad page geometry (page_size=%lld, max_pages_per_seq=%lld)", (long long)page_size,
                (long long)max_pages_per_seq);
  hipStream_t st = (hipStream_t)stream;
  VSEL_PROF_BEGIN(st);
  return attn_launch(st, q, k_cache, v_cache, cu_seqlens_q, n_seq, max_seqlen_q, hq, hkv, d, scale, causal, out,
                     PagedKV{seqlens_k, block_table, nullptr, (int)max_pages_per_seq, (int)page_size, 0, 0, 0, 0, 0, 0});
}
