// Var-len causal GQA attention forward (placeholder until the MFMA kernel lands in this round).
#include "common.h"
using namespace vsel;
extern "C" int vsel_varlen_attn_fwd(void* stream, const void* q, const void* k, const void* v, const int32_t* cu_seqlens,
                                    int64_t n_seq, int64_t max_seqlen, int64_t total, int64_t hq, int64_t hkv, int64_t d,
                                    float scale, int causal, void* out) {
  return fail(VSEL_ERR_UNSUPPORTED, "vsel_varlen_attn_fwd: kernel not built yet");
}
