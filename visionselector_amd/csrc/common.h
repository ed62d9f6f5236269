// Shared device/host helpers for libvsel (gfx950 / CDNA4 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <type_traits>

#include "../../include/vsel.h"
#include "../../include/vsel_debug.h"

namespace vsel {

constexpr int kWave = 64;

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

struct bf16_t { uint16_t bits; };

// -DVSEL_TRACE (tools/trace_small.py / tools/trace_attn.py build with it; never in the shipped library): thread 0 of every
// workgroup leaves s_memrealtime stamps (100 MHz, one clock for the whole device) at phase edges.  One table per translation unit.
#ifdef VSEL_TRACE
constexpr int kTraceKernels = 8, kTraceBlocks = 1024, kTraceSlots = 8;
static __device__ unsigned long long g_trace[kTraceKernels][kTraceBlocks][kTraceSlots];
#define VSEL_STAMP(kern, slot)                                                                                         \
  do {                                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    if (threadIdx.x == 0) {                                                                                            \
      const unsigned b_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;                              \
      if (b_ < (unsigned)kTraceBlocks) g_trace[kern][b_][slot] = __builtin_amdgcn_s_memrealtime();                     \
    }                                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
  } while (0)
#define VSEL_STAMP_DRAIN(kern, slot)                                                                                   \
  do {                                                                                                                 \
    __builtin_amdgcn_s_waitcnt(0);                                                                                     \
    VSEL_STAMP(kern, slot);                                                                                            \
  } while (0)
#else
#define VSEL_STAMP(kern, slot) do {} while (0)
#define VSEL_STAMP_DRAIN(kern, slot) do {} while (0)
#endif

// ---- diagnostic knobs (include/vsel_debug.h): one process-global table, relaxed atomic reads -------------------------
int knob(int id);
// XCD-local work queues of the attention kernels (attn_common.h, XcdQueue): knob VSEL_KNOB_ATTN_XCD_QUEUE forces them on / off;
// -1 (default) = on from min_len tokens in the longest sequence and min_pairs (sequence, kv head) pairs.  The thresholds are per
// kernel (same-binary A/B on MI355X, profiles/r04_attn_xcd_queue.txt): a pair-major list is a worse load balance than heaviest-
// first over all pairs when a pair has few items, so short / few sequences keep the single queue.
int attn_use_xcd_queues(int64_t max_seqlen, int64_t n_pairs, int64_t min_len, int64_t min_pairs);
// static deal of the work items instead of the atomic queue (attn_common.h::static_deal_item; knob attn_static): forced by the knob, else
// when `auto_ok` and the grid has few rounds of items: n_items <= rounds20 / 20 slots (2.35 rounds for the 32-rows-per-wave kernels, 1.8
// for the 64-rows forms, whose ragged batches balance worse: tools/exp_attn_static_all.py, profiles/r04_attn_static.txt)
bool attn_static_deal(int64_t n_items, int64_t slots, bool auto_ok, int rounds20 = 47);

// ---- work-queue counter slots of the persistent attention kernels ---------------------------------------------------------------
// Every queued launch zeroes and then drains a small device counter; each kernel family owns 64 of them ("slots").  A slot belongs to a
// STREAM: launches of one stream reuse that stream's slot (stream order protects the counter), a new stream takes a free slot, and when
// 64 other streams hold them all a slot is taken over from a stream with nothing in flight (hipStreamQuery) -- else the call fails with
// VSEL_ERR_BUSY instead of letting two concurrent launches share a counter.  Not covered: a captured graph bakes in the slot of its
// capture stream, so it must not be replayed concurrently with itself or with eager launches on that stream.
enum QueueFamily { kSlotFwd = 0, kSlotFwd64, kSlotGqa, kSlotBwd, kSlotDq64, kSlotDkdv64, kSlotFamilies };
int queue_slot_acquire(int family, hipStream_t st, int* slot);       // VSEL_OK, VSEL_ERR_BUSY, VSEL_ERR_HIP
void queue_slot_launched(int family, int slot, hipStream_t st);      // after the launch that uses `slot` has been enqueued on st

// ---- error plumbing --------------------------------------------------------------------------
void set_error(const std::string& msg);
int fail(vsel_status st, const char* fmt, ...);
const std::string& last_error();

#define VSEL_HIP_CHECK(expr)                                                              \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) return ::vsel::fail(VSEL_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

#define VSEL_LAUNCH_CHECK(name)                                                           \
  do {                                                                                    \
    hipError_t _e = hipGetLastError();                                                    \
    if (_e != hipSuccess) return ::vsel::fail(VSEL_ERR_HIP, "launch %s: %s", name, hipGetErrorString(_e)); \
  } while (0)

// ---- optional per-kernel timing (vsel_profile_start/stop): HIP events on the launch stream -------
bool prof_enabled();
void prof_mark(hipStream_t st, const char* name);   // event AFTER the named kernel ("<begin>" opens an API call)

#define VSEL_AFTER_LAUNCH(st, name)                                                       \
  do {                                                                                    \
    VSEL_LAUNCH_CHECK(name);                                                              \
    if (::vsel::prof_enabled()) ::vsel::prof_mark(st, name);                              \
  } while (0)
#define VSEL_PROF_BEGIN(st)                                                               \
  do {                                                                                    \
    if (::vsel::prof_enabled()) ::vsel::prof_mark((hipStream_t)(st), "<begin>");          \
  } while (0)
// Every kernel launch of the library goes through this: while the profiler is on, a "<begin>" event is recorded on the launch
// stream right in front of the kernel, so that the interval booked to the kernel (begin mark -> the mark VSEL_AFTER_LAUNCH records)
// holds the kernel and two marker packets, NOT the host time that passed since the previous launch was enqueued.
#define VSEL_LAUNCH(kernel, grid, block, lds_bytes, st, ...)                              \
  do {                                                                                    \
    if (::vsel::prof_enabled()) ::vsel::prof_mark((hipStream_t)(st), "<begin>");          \
    hipLaunchKernelGGL(kernel, grid, block, lds_bytes, st, __VA_ARGS__);                  \
  } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- element access: 16-byte vectors ----------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<bf16_t> {
  static constexpr int kVec = 8;   // elements per 16-byte lane load
};
template <> struct Elem<float> {
  static constexpr int kVec = 4;
};

__device__ __forceinline__ float bf16_to_f32(uint32_t bits16) { return __uint_as_float(bits16 << 16); }

// round-to-nearest-even fp32 -> bf16 (matches torch .to(bfloat16)); NaN stays NaN
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0u;
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}

// load kVec consecutive elements (16 B, must be 16-B aligned) and widen to fp32
__device__ __forceinline__ void load_vec(const bf16_t* p, float (&v)[8]) {
  const u32x4 r = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(r[i] << 16);
    v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ void load_vec(const float* p, float (&v)[4]) {
  const f32x4 r = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = r[i];
}
// streaming variants: the token tensor is far larger than the caches and each sweep reads a byte once -> non-temporal loads
__device__ __forceinline__ void load_vec_stream(const bf16_t* p, float (&v)[8]) {
  const u32x4 r = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(r[i] << 16);
    v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ void load_vec_stream(const float* p, float (&v)[4]) {
  const f32x4 r = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = r[i];
}
__device__ __forceinline__ void store_vec(bf16_t* p, const float (&v)[8]) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = f32_to_bf16_bits(v[2 * i]) | (f32_to_bf16_bits(v[2 * i + 1]) << 16);
  *reinterpret_cast<u32x4*>(p) = r;
}
__device__ __forceinline__ void store_vec(float* p, const float (&v)[4]) {
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = v[i];
  *reinterpret_cast<f32x4*>(p) = r;
}
__device__ __forceinline__ float load_elem(const bf16_t* p) { return bf16_to_f32(p->bits); }
__device__ __forceinline__ float load_elem(const float* p) { return *p; }
__device__ __forceinline__ void store_elem(bf16_t* p, float v) { p->bits = (uint16_t)f32_to_bf16_bits(v); }
__device__ __forceinline__ void store_elem(float* p, float v) { *p = v; }

// ---- wave64 reductions (fixed order => deterministic) -----------------------------------------
// DPP instead of __shfl_xor: hipcc lowers every __shfl_xor to ds_bpermute_b32 (an LDS-crossbar round trip, ~6 dependent
// ones per reduction); quad_perm / row_mirror DPP modifiers ride on the VALU op itself.  After 4 DPP steps every lane
// holds its 16-lane row's result; the 4 rows are combined through v_readlane (uniform values).
template <typename Op>
__device__ __forceinline__ float wave_reduce_dpp(float v, Op op) {
  auto dpp = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false));
  };
  v = op(v, dpp(v, std::integral_constant<int, 0xB1>{}));    // quad_perm [1,0,3,2]
  v = op(v, dpp(v, std::integral_constant<int, 0x4E>{}));    // quad_perm [2,3,0,1]
  v = op(v, dpp(v, std::integral_constant<int, 0x141>{}));   // row_half_mirror
  v = op(v, dpp(v, std::integral_constant<int, 0x140>{}));   // row_mirror
  const int vi = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
  return op(op(r0, r1), op(r2, r3));
}
// inclusive prefix sum over the 64 lanes (integer => exact): 4 row_shr DPP steps inside each 16-lane row, then the totals of
// the lower rows through v_readlane.  (__shfl_up / __shfl_down cost one ds_bpermute_b32 round trip per step.)
__device__ __forceinline__ uint32_t wave_prefix_sum_u32(uint32_t v) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);    // row_shr:1 (lanes without a source add 0)
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);    // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);    // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);    // row_shr:8
  const int t0 = __builtin_amdgcn_readlane(x, 15), t1 = __builtin_amdgcn_readlane(x, 31), t2 = __builtin_amdgcn_readlane(x, 47);
  const int lane = (int)(threadIdx.x & 63);
  x += (lane >= 16 ? t0 : 0) + (lane >= 32 ? t1 : 0) + (lane >= 48 ? t2 : 0);
  return (uint32_t)x;
}
__device__ __forceinline__ float wave_sum(float v) { return wave_reduce_dpp(v, [](float a, float b) { return a + b; }); }
__device__ __forceinline__ float wave_max(float v) { return wave_reduce_dpp(v, [](float a, float b) { return fmaxf(a, b); }); }
__device__ __forceinline__ float wave_min(float v) { return wave_reduce_dpp(v, [](float a, float b) { return fminf(a, b); }); }

// ---- segments -----------------------------------------------------------------------------------
struct SegView {
  const int32_t* seg_rows;  // device or null
  const int32_t* seg_out;   // device or null
  int32_t rows_per_seg;
  int32_t k;
  __device__ __forceinline__ int64_t row_begin(int s) const { return seg_rows ? (int64_t)seg_rows[s] : (int64_t)s * rows_per_seg; }
  __device__ __forceinline__ int32_t n_rows(int s) const { return seg_rows ? seg_rows[s + 1] - seg_rows[s] : rows_per_seg; }
  __device__ __forceinline__ int64_t out_begin(int s) const { return seg_out ? (int64_t)seg_out[s] : (int64_t)s * k; }
  __device__ __forceinline__ int32_t n_out(int s) const { return seg_out ? seg_out[s + 1] - seg_out[s] : k; }
};

inline SegView make_view(const vsel_segments* seg) {
  SegView v;
  v.seg_rows = seg->seg_rows;
  v.seg_out = seg->seg_out;
  v.rows_per_seg = (int32_t)seg->rows_per_seg;
  v.k = (int32_t)seg->k;
  return v;
}

int check_segments(const vsel_segments* seg, bool need_k);

}  // namespace vsel
