// Internal interface between attn.hip (dispatch of lgd_attn_fwd_f16) and attn_w4.hip (the d = 40 self-attention
// forward kernel of round 4).  Not part of the C ABI.
#pragma once
#include "common.h"

struct AttnW4Args {
  const half_t* q; long ldq, q_bs;
  const half_t* k; long ldk, k_bs;
  const half_t* v; long ldv, v_bs;
  half_t* o; long ldo, o_bs;
  float* lse;               // optional: log2-domain log-sum-exp [B][H][Sq]
  int B, H, Sq, Sk, d;
  float scale_log2;         // scale * log2(e)
};

int lgd_attn_w4_supported(const AttnW4Args& a);
int lgd_attn_w4_launch(const AttnW4Args& a, hipStream_t st);
void lgd_attn_w4_set_pipe(int v);          // 1: one wave per SIMD, in-wave software pipeline; 0: two waves per SIMD
