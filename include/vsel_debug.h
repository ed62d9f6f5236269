/* vsel_debug.h -- diagnostic knobs of libvsel.so (NOT part of the drop-in boundary; include/vsel.h is).
 *
 * The library picks its kernel forms (small-batch LIS path, fused select, two-stream pipeline, attention workgroup shapes,
 * backward split) from the problem size.  The -m gpu tests and the tools/ benchmarks need to FORCE a form to prove that the
 * forms agree bit for bit and to measure them against each other; these knobs are how.  They replace nothing in the reference.
 *
 * Contract
 *   - one table of process-global ints, read with relaxed atomics at launch time: setting a knob is thread-safe, but a call
 *     running concurrently on another thread may see either value (one process per GPU, no intra-process threading is the
 *     supported model);
 *   - vsel_debug_set returns the PREVIOUS value through *previous so that a caller can restore it (the Python side wraps this
 *     in a context manager: visionselector_amd._native.debug_knob); vsel_debug_reset puts every knob back to its default
 *     (the default honours the environment variable named below, read once);
 *   - results never depend on a knob beyond what the knob's line says (every form is parity-tested against the same oracle).
 */
#ifndef VSEL_DEBUG_H_
#define VSEL_DEBUG_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef enum vsel_debug_knob {
  VSEL_KNOB_LIS_PIPELINE = 0,     /* 0 / 1: two-half aux-stream pipeline for >= 32 segments (env VSEL_PIPELINE, default 0); bit-identical */
  VSEL_KNOB_LIS_SMALL_PATH = 1,   /* small-batch LIS form up to this many segments (env VSEL_SMALL_PATH, default 4, 0 = never); bit-identical */
  VSEL_KNOB_LIS_FUSED_SELECT = 2, /* radix select fused into the gather up to this many segments (env VSEL_FUSED_SELECT, default 48); bit-identical */
  VSEL_KNOB_ATTN_USE_TR = 3,      /* 0 / 1: V fragments by ds_read_b64_tr_b16 (default 1) or by plain LDS reads; bit-identical */
  VSEL_KNOB_ATTN_WAVES = 4,       /* 0 = by grid size (default), 4 / 8 = force the forward workgroup size; bit-identical */
  VSEL_KNOB_ATTN_PACK = 5,        /* GQA-packed decode form: 0 never, 1 whenever it applies, 2 (default) by grid size; bit-identical */
  VSEL_KNOB_ATTN_SPLIT = 6,       /* two-KV-stream form for one short sequence: 0 never, 1 whenever <= 256 items, 2 (default) ... and
                                     the sequences are not tiny; inference only, sums in a different order (bf16-rounding agreement) */
  VSEL_KNOB_ATTN_SPLIT_Q64 = 7,   /* 0 / 1: 64-query workgroups in the two-stream form when they fit one per CU (default 1) */
  VSEL_KNOB_ATTN_BWD_SPLIT = 8,   /* dK/dV items: -1 by item count, sequence length and raggedness (default), 0 the group's q heads inside the
                                     item, 1 one q head per item, 2 / 3 / 4 the group in that many parts (not in the 4-wave kernel: 1 there);
                                     fp32 partials + ordered group sum in forms >= 1; deterministic in every form, fp32 association
                                     differs between them (dK / dV agree to a bf16 rounding) */
  VSEL_KNOB_LIS_SPLICE_FUSED = 9, /* 0 / 1: vsel_lis_select_splice writes kept rows straight into inputs_embeds' (default 1) or runs
                                     select then splice as two steps; bit-identical */
  VSEL_KNOB_ATTN_BWD_WAVES = 10,  /* 4 / 8: dK/dV workgroup: four waves with K/V operands in registers, or eight (two per SIMD, query tile split
                                     across wave pairs, K/V fragments from LDS); deterministic either way, fp32 association differs */
  VSEL_KNOB_ATTN_TAIL_FIRST = 11, /* query tiles aligned to the END of each sequence (the partial tile is the cheap first one): -1 when
                                     the grid is throughput-bound (default), 0 / 1 force; bit-identical outputs */
  VSEL_KNOB_LIS_SEG_SUMS = 12,    /* sweep 1 by one wave per (segment, column slab), no per-chunk partials: the widest slab (512 / 256 / 128
                                     bf16 columns) that gives this many pairs, else the chunked form (default 896, 0 = never; env
                                     VSEL_SEG_SUMS); bit-identical sums */
  VSEL_KNOB_ATTN_XCD_QUEUE = 13,  /* attention work items on XCD-local queues, one (sequence, kv head) pair's items on one XCD so that the
                                     stream they share stays in that XCD's L2: -1 by sequence length and pair count per kernel (default), 0 / 1 force (env
                                     VSEL_ATTN_XCD_QUEUE); placement only, outputs bit-identical */
  VSEL_KNOB_ATTN_ROWS64 = 14,     /* forward for head_dim 128 by the 64-rows-per-wave kernel (one 512-register wave per SIMD, hand-scheduled
                                     tile loop; csrc/attn_fwd64.hip): -1 from 2048 tokens in the longest sequence (default), 0 never,
                                     1 whenever it applies (contiguous keys); env VSEL_ATTN_ROWS64; bit-identical outputs */
  VSEL_KNOB_ATTN_BWD_DQ64 = 15,   /* dQ pass of the attention backward by the 64-rows-per-wave kernel (csrc/attn_bwd_dq64.hip, generated body):
                                     -1 from 1024 tokens in the longest sequence (default), 0 never, 1 always; env VSEL_ATTN_BWD_DQ64; dQ / D / lse2
                                     bit-identical */
  VSEL_KNOB_ATTN_BWD_DKDV64 = 16, /* dK / dV pass by the one-wave-per-SIMD kernel with the unit pipeline (csrc/attn_bwd_dkdv64.hip, generated
                                     body; either item form): -1 from 1024 tokens in the longest sequence, 2048 in the per-q-head split form
                                     (default), 0 never, 1 always; env VSEL_ATTN_BWD_DKDV64; dK / dV as the 4-wave form's, bit for bit */
  VSEL_KNOB_ATTN_STATIC = 17,     /* forward, 4- / 8-wave workgroups: work items dealt out statically (workgroup b takes items b, 2 G - 1 - b,
                                     2 G + b, ... of the heaviest-first list: no atomic, no hand-over barriers, the next item known in advance):
                                     -1 by item count and sequence length (default; the group-shared forms attn_fwd_gqa / attn_fwd_gqa64 deal
                                     every uniform batch and draw ragged ones behind two dealt rounds), 0 / 1 force; env VSEL_ATTN_STATIC;
                                     placement only, outputs bit-identical */
  VSEL_KNOB_ATTN_SKIP_EMPTY = 18, /* single work queue on ragged batches: the shared counter jumps over runs of EMPTY items (levels a shorter
                                     sequence does not reach) instead of handing each one out: 1 (default) / 0; env VSEL_ATTN_SKIP_EMPTY;
                                     placement only, outputs bit-identical */
  VSEL_KNOB_ATTN_GQA = 19,        /* forward for head_dim 128, prefill over the queries' own keys, 2 <= hq / hkv <= 8: ONE 8-wave workgroup serves a
                                     kv head's whole q-head group on a 32-query tile (K / V tiles loaded once per group, 32-key causal granularity,
                                     the next item's rows in flight under the last tile; csrc/attn_fwd_gqa.hip): -1 for throughput-bound grids of
                                     sequences below 2048 tokens (default), 0 never, 1 whenever it applies; env VSEL_ATTN_GQA; bit-identical outputs */
  VSEL_KNOB_ATTN_GQA_FORM = 20,   /* which group-shared forward: -1 by item count and group size (default), 0 the 8-wave form (one head per wave,
                                     items pipelined into each other; csrc/attn_fwd_gqa.hip), 1 the generated 64-rows-per-wave loop with two heads
                                     per wave (csrc/attn_fwd_gqa64.hip); env VSEL_ATTN_GQA_FORM; bit-identical outputs */
  VSEL_KNOB_ATTN_BWD_UPDOWN = 21, /* dK / dV pass, causal, even number of kv heads: items of odd kv heads walk their query tiles UPWARD from the item's
                                     first query (even ones downward from the sequence's end) and (sequence, kv head) pairs are queued in couples
                                     per XCD, so the two directions alternate on a queue and its CUs stay on one Q / dO tile per pair (L2);
                                     default 1; env VSEL_ATTN_BWD_UPDOWN; a rule of the item alone (batch-invariant), fp32 association of odd
                                     heads differs from the 0 setting */
  VSEL_KNOB_ATTN_KEY_PARTS = 22,  /* vsel_varlen_attn_fwd_ws: key-range parts of the 256-query items (csrc/attn_fwd64_parts.hip): -1 for uniform causal batches
                                     of <= 128 items (few q heads) from 2048 tokens (default), 0 never, 1 whenever the shapes allow, n > 1: n key tiles per part;
                                     env VSEL_ATTN_KEY_PARTS; deterministic, another fp32 association than the unsplit forms */
  VSEL_KNOB_LIS_GATHER = 23,      /* the gather of the kept rows for uniform segments: 0 = one workgroup per 4 .. 32 kept rows of a segment, > 0 = a flat
                                     list of kept rows dealt to resident waves, value = 10 * workgroups per CU + rows in flight per wave (2 .. 4);
                                     env VSEL_GATHER; bit-identical (a copy) */
  VSEL_KNOB_TRAIN_FUSED = 24,     /* 0 / 1: training backward of one row of <= 4096 scores (bf16 weights) in five launches -- the soft top-k backward in the
                                     prologue of the weighted column sums, both projections in one launch, one finish kernel, both rank-1 writes in one
                                     launch -- instead of ten (default 1; env VSEL_TRAIN_FUSED); bit-identical */
  VSEL_KNOB_COUNT = 25
} vsel_debug_knob;

/* Set knob to value (clamped to the knob's range).  *previous (may be NULL) receives the value it had.
 * Returns 0, or 1 (VSEL_ERR_INVALID) for an unknown knob. */
int vsel_debug_set(int knob, int value, int* previous);
/* Current value of knob through *value.  Returns 0 or 1 as above. */
int vsel_debug_get(int knob, int* value);
/* Every knob back to its default. */
void vsel_debug_reset(void);

#ifdef VSEL_TRACE
/* Only in libraries built with -DVSEL_TRACE (tools/trace_small.py, tools/trace_attn.py; never the shipped one): copy the
 * s_memrealtime stamps of the last small-batch LIS call / attention forward launch,
 * out[kTraceKernels = 8][kTraceBlocks = 1024][kTraceSlots = 8] uint64; clear != 0 zeroes the device table afterwards. */
int vsel_debug_read_trace(unsigned long long* out, int clear);
int vsel_debug_read_attn_trace(unsigned long long* out, int clear);
/* s_memtime stamps of one steady-state dK / dV tile of workgroup 0: out[8 waves][8 slots] (tools/trace_attn_bwd.py) */
int vsel_debug_read_bwd_trace(unsigned long long* out);
/* the same for one steady-state tile of the attention forward's main loop: out[8 waves][8 slots] (tools/trace_attn_fwd.py) */
int vsel_debug_read_fwd_tile_trace(unsigned long long* out);
#endif

#ifdef __cplusplus
}
#endif
#endif /* VSEL_DEBUG_H_ */
