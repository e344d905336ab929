// Whole relation-scoring pass drivers (gfx950): they only sequence the kernels
// of this library on the caller's stream — no allocation, no synchronisation.
//
// tspn_forward_fused_f32: the product path on tracklet tensors.  The k=3
// temporal conv of DPNHead (reference lib/modeling/relpn/dpn.py:70) is linear
// in its input channels, and the pair feature is the channel concatenation
// (subject ‖ object), so
//     conv(cat(f_s, f_o)) = W[:, :D] * f_s  +  W[:, D:] * f_o
// The N per-tracklet projections U = W_s*f + bias and V = W_o*f are computed
// once (one MFMA implicit GEMM with M = 2C) instead of N(N-1) times, and the
// pair stage forms relu(U[s] + V[o]) in registers on the way into the heads'
// MFMAs.  The N^2 x C x T pair tensor and its ReLU image never touch HBM.
//
// tspn_temporal_encoder_heads_f32: the reference-faithful dense form on an
// arbitrary materialised [P, C, T] (what DPNHead.forward would be handed).
#include <algorithm>

#include "tspn_common.h"

namespace {

struct FusedLayout {
  size_t xt, y, bias2, fbar, pooled, lin, vt, hwp, hot, total;
  size_t vt_bytes;
  size_t lin_bytes;
};

FusedLayout layout_of(const tspn_fused_desc* d) {
  FusedLayout L{};
  const size_t NT = (size_t)d->B * d->N, T = d->T, D = d->D, C = 2 * D;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off += tspn::align_up(bytes, 256);
    return o;
  };
  L.xt = take(NT * D * T * sizeof(float));
  // rows of y may be padded to a multiple of 4 frames (ldy), see tspn_forward_fused_f32
  L.y = take(NT * 2 * C * tspn::align_up(T, 4) * sizeof(float));
  L.bias2 = take(2 * C * sizeof(float));
  L.fbar = take(NT * D * sizeof(float));
  L.pooled = take(256);  // (unused since the predicate head is evaluated per tracklet)
  L.lin_bytes = tspn::pair_predicate_workspace_bytes((int64_t)NT, (int64_t)D, d->K);
  L.lin = take(L.lin_bytes);
  // Winograd-transformed input V of the F(6,3) kernel (conv_algo TSPN_CONV_WINOGRAD63)
  L.vt_bytes = (D % 32 == 0) ? tspn::wino63_workspace_bytes((int64_t)NT, (int64_t)T, (int64_t)D) : 0;
  L.vt = take(L.vt_bytes);
  L.hwp = take(C * 12 * sizeof(float));   // head weights packed [C][12] for the scalar-weight pair stage (H == 12)
  L.hot = take(TSPN_CONV_CHECK_SCRATCH_BYTES);   // accuracy guard: scratch of tspn_conv3_spot_check_f32 (tspn_conv_guard.hip)
  L.total = off;
  return L;
}

int check_desc(const tspn_fused_desc* d) {
  TSPN_REQUIRE(d != nullptr, TSPN_EINVAL, "tspn_forward_fused: null descriptor");
  TSPN_REQUIRE(d->B >= 0 && d->N >= 0 && d->T > 0 && d->D > 0 && d->A > 0 && d->K > 0 &&
                   d->P >= 0,
               TSPN_EINVAL, "tspn_forward_fused: bad sizes B=%lld N=%lld T=%lld D=%lld A=%lld K=%lld P=%lld",
               (long long)d->B, (long long)d->N, (long long)d->T, (long long)d->D, (long long)d->A,
               (long long)d->K, (long long)d->P);
  TSPN_REQUIRE(3 * d->A <= 16, TSPN_EUNSUPPORTED, "tspn_forward_fused: 3*A=%lld > 16",
               (long long)(3 * d->A));
  return TSPN_OK;
}

}  // namespace

extern "C" size_t tspn_forward_fused_workspace_bytes(const tspn_fused_desc* d) {
  if (check_desc(d) != TSPN_OK) return 0;
  return layout_of(d).total;
}

extern "C" int tspn_forward_fused_f32(const tspn_fused_desc* d, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (d->P == 0 || d->B * d->N == 0) return TSPN_OK;
  TSPN_REQUIRE(d->feats && d->pairs && d->conv_packed && d->conv_bias && d->head_w && d->head_b &&
                   d->cls_w && d->cls_b && d->out_heads && d->out_logits,
               TSPN_EINVAL, "tspn_forward_fused: null pointer in descriptor");
  const FusedLayout L = layout_of(d);
  TSPN_REQUIRE(d->workspace && d->workspace_bytes >= L.total, TSPN_EWORKSPACE,
               "tspn_forward_fused: workspace %zu < %zu bytes", d->workspace_bytes, L.total);
  TSPN_REQUIRE((reinterpret_cast<uintptr_t>(d->workspace) & 255) == 0, TSPN_EINVAL,
               "tspn_forward_fused: workspace must be 256-byte aligned");
  char* ws = static_cast<char*>(d->workspace);
  float* xt = reinterpret_cast<float*>(ws + L.xt);
  float* y = reinterpret_cast<float*>(ws + L.y);
  float* bias2 = reinterpret_cast<float*>(ws + L.bias2);
  float* fbar = reinterpret_cast<float*>(ws + L.fbar);
  float* pooled = reinterpret_cast<float*>(ws + L.pooled);
  void* lin = ws + L.lin;
  const int64_t NT = d->B * d->N, T = d->T, D = d->D, C = 2 * D, H = 3 * d->A;
  hipStream_t s = TSPN_STREAM(stream);

  // bias of the encoder goes with the subject projection: bias2 = [conv_bias, 0]
  hipError_t e = hipMemsetAsync(bias2 + C, 0, C * sizeof(float), s);
  if (e == hipSuccess)
    e = hipMemcpyAsync(bias2, d->conv_bias, C * sizeof(float), hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess)
    return tspn::fail(TSPN_ELAUNCH, "tspn_forward_fused: bias staging: %s", hipGetErrorString(e));

  // 0. RelOIPool over the segment on the pair feats (= cat of the two tracklet means) + predicate head;
  // linear in the two halves, so evaluated per tracklet and combined per pair.  Done FIRST: the logits do not
  // depend on the encoder, and a caller may decode / gather them on another stream behind ev_logits_ready.
  if ((rc = tspn_temporal_mean_f32(d->feats, NT, T, D, 1, fbar, stream))) return rc;
  (void)pooled;
  if ((rc = tspn::pair_predicate(fbar, NT, D, d->pairs, d->P, d->cls_w, d->cls_b, d->K, d->out_logits, lin,
                                 L.lin_bytes, stream)))
    return rc;
  if (d->ev_logits_ready) (void)hipEventRecord(static_cast<hipEvent_t>(d->ev_logits_ready), s);

  // 1+2. per-tracklet projections  y[NT, 2C, T]: rows [0,C) = U (+bias), rows [C,2C) = V.
  // The channels-last kernel consumes the tracklet layout [NT,T,D] directly; ragged shapes go
  // through a transpose to channels-first [NT,D,T] and the general kernel.
  const bool tc = (D % 16 == 0) && ((reinterpret_cast<uintptr_t>(d->feats) & 15) == 0) &&
                  ((reinterpret_cast<uintptr_t>(d->conv_packed) & 15) == 0);
  TSPN_REQUIRE(d->conv_algo == TSPN_CONV_DIRECT || d->conv_algo == TSPN_CONV_WINOGRAD63, TSPN_EINVAL,
               "tspn_forward_fused: conv_algo must be TSPN_CONV_DIRECT (0) or TSPN_CONV_WINOGRAD63 (1), got %d",
               d->conv_algo);
  const bool w63 = d->conv_algo == TSPN_CONV_WINOGRAD63;
  TSPN_REQUIRE(!w63 || (tc && tspn::wino63_supported(D, 2 * C)), TSPN_EUNSUPPORTED,
               "tspn_forward_fused: TSPN_CONV_WINOGRAD63 needs D %% 32 == 0 and 16-byte aligned operands "
               "(pack the weights with tspn_pack_conv3_f32 and pass TSPN_CONV_DIRECT otherwise)");
  // On the fast path the rows of y are padded to ldy = ceil4(T) frames so that the blocked pair stage
  // can stage them with 16-byte LDS-DMA pieces that never leave a row (pad frames are never read out).
  // (needs what the DMA pair-stage kernel needs: even T, C % 16 == 0 — implied by tc)
  const int64_t ldy =
      (tc && d->canonical_pairs && T % 2 == 0) ? (int64_t)tspn::align_up((size_t)T, 4) : T;
  if (!tc && (rc = tspn_transpose_td_f32(d->feats, NT, T, D, xt, stream))) return rc;
  // F(6,3): the Winograd input transform runs as its own HBM-bound pass; the profiling events bracket the
  // MFMA kernel only
  // Accuracy guard of F(6,3) (d->conv_check rows, needs the raw weights and an attached status block): the transform
  // reports the sextet with the largest |x|; behind the conv a few outputs are recomputed in float64 (tspn_conv_guard.hip)
  const bool guard = w63 && d->conv_check > 0 && d->conv_weight != nullptr;
  uint64_t* hot = guard ? reinterpret_cast<uint64_t*>(ws + L.hot) : nullptr;
  // (scratch of the spot check: the workgroups' meeting point + the slots the transform reports the hot sextet into)
  if (guard && hipMemsetAsync(hot, 0, TSPN_CONV_CHECK_SCRATCH_BYTES, s) != hipSuccess)
    return tspn::fail(TSPN_ELAUNCH, "tspn_forward_fused: guard staging: %s", hipGetErrorString(hipGetLastError()));
  if (w63 && (rc = tspn::wino63_input_transform(d->feats, NT, T, D, ws + L.vt, L.vt_bytes, stream,
                                                hot ? hot + TSPN_CONV_CHECK_HOT_OFFSET / 8 : nullptr)))
    return rc;
  if (d->ev_conv_begin) (void)hipEventRecord(static_cast<hipEvent_t>(d->ev_conv_begin), s);
  if (w63)
    rc = tspn::wino63_contract(ws + L.vt, NT, T, D, d->conv_packed, 2 * C, bias2, 0, y, ldy, stream);
  else
    rc = tc ? tspn::conv3_tc_direct(d->feats, NT, T, D, d->conv_packed, 2 * C, bias2, 0, y, ldy, stream)
            : tspn_conv3_f32(xt, NT, D, T, d->conv_packed, 2 * C, bias2, 0, y, stream);
  if (rc) return rc;
  if (d->ev_conv_end) (void)hipEventRecord(static_cast<hipEvent_t>(d->ev_conv_end), s);
  if (guard && (rc = tspn_conv3_spot_check_f32(d->feats, NT, T, D, d->conv_weight, C, C, D, bias2, 0, y, ldy, hot,
                                               d->conv_check, stream)))
    return rc;
  // 3. pair stage + relationness / span heads
  if (d->canonical_pairs) {
    TSPN_REQUIRE(d->P == d->B * d->N * (d->N - 1), TSPN_EINVAL,
                 "tspn_forward_fused: canonical_pairs needs P == B*N*(N-1) (P=%lld)", (long long)d->P);
    if ((rc = tspn::heads_pairgrid(y, ldy, d->B, d->N, C, T, d->head_w, d->head_b, H, d->out_heads,
                                   stream, reinterpret_cast<float*>(ws + L.hwp))))
      return rc;
  } else if ((rc = tspn_heads_f32(1, y, y + C * T, 2 * C, d->pairs, d->pairs + 1, 2, nullptr,
                                  d->head_w, d->head_b, H, d->P, C, T, d->out_heads, stream))) {
    return rc;
  }
  return TSPN_OK;
}

extern "C" int tspn_temporal_encoder_heads_f32(const float* x, int64_t P, int64_t C, int64_t T,
                                               const float* conv_packed, const float* conv_bias,
                                               const float* head_w, const float* head_b, int64_t H,
                                               float* h_ws, float* out_heads, void* stream) {
  if (P == 0) return TSPN_OK;
  TSPN_REQUIRE(x && conv_packed && head_w && h_ws && out_heads, TSPN_EINVAL,
               "tspn_temporal_encoder_heads_f32: null pointer");
  int rc = tspn_conv3_f32(x, P, C, T, conv_packed, C, conv_bias, 1, h_ws, stream);
  if (rc) return rc;
  return tspn_heads_f32(0, h_ws, nullptr, C, nullptr, nullptr, 1, nullptr, head_w, head_b, H, P, C,
                        T, out_heads, stream);
}
