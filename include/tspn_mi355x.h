/*
 * tspn_mi355x.h — C ABI of the MI355X-native TSPN relation-scoring hot path.
 *
 * The reference (sangminwoo/Temporal-Span-Proposal-Network-VidVRD) is pure
 * Python and has no FFI / operator layer: the seam it offers is the
 * `BaseModel.forward()` nn.Module API (lib/modeling/model.py:20-24).  This
 * header is the native boundary *under* that API: each entry point replaces the
 * torch-op sequence cited next to it.  The Python host
 * (temporal-span-proposal-network-vidvrd_amd/_abi.py) binds these symbols with
 * ctypes; INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - All pointers are raw DEVICE pointers (HBM) owned by the caller (torch's
 *     allocator); the library borrows them for the duration of the call and
 *     allocates nothing.  Scratch space is passed in explicitly (see the
 *     *_workspace_bytes helpers).
 *   - All tensors are dense, row-major, fp32 unless stated; sizes are int64.
 *   - `stream` is a hipStream_t passed as void* (torch's current stream).  The
 *     functions enqueue work and return; they never synchronise, never touch
 *     the null stream implicitly, and are re-entrant.
 *   - Return value: 0 on success, negative TSPN_E* on failure;
 *     tspn_last_error() returns a thread-local message.  No C++ exception
 *     crosses this boundary.
 *   - Target: gfx950 only.
 */
#ifndef TSPN_MI355X_H
#define TSPN_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct layout or a signature changes (2: tspn_fused_desc gained conv_algo /
 * canonical_pairs in round 1 without a bump; round 2 adds the struct-size exports below, which the
 * host checks against its own view of the descriptors at load time; 3: tspn_fused_desc.ev_logits_ready;
 * 4 (round 3): the Winograd F(2,3) and the three F(4,3) temporal-conv generations and their pack / repack
 * entry points are gone, tspn_fused_desc.conv_algo is TSPN_CONV_DIRECT | TSPN_CONV_WINOGRAD63;
 * 5 (round 3): tspn_pack_conv2d_frag_bf16 lays the fragments out channel chunk by chunk (was tap by tap) and
 * the bf16 convolutions contract in that order; tspn_stem_pool_bf16 added);
 * 6 (round 5, additive): tspn_bottleneck_block_bf16, tspn_bottleneck_block_proj_bf16, tspn_bottleneck_block_res_bf16,
 * tspn_bottleneck_tail_io_bf16, tspn_conv3_tc_wino63_set_piece_form (replaces the TSPN_WINO63_PTRV environment switch);
 * 7 (round 6): the device status block (tspn_status_attach / _fault / _clear / _selftest, TSPN_EDEVICE: a kernel can
 * raise a fault that the NEXT launch entry reports without a synchronisation); tspn_fused_desc gained conv_weight /
 * conv_check at its END (the a-posteriori accuracy guard of the F(6,3) temporal conv, tspn_conv3_spot_check_f32). */
#define TSPN_ABI_VERSION 7

enum {
  TSPN_OK = 0,
  TSPN_EINVAL = -1,       /* bad argument (null pointer, negative size, misalignment) */
  TSPN_EUNSUPPORTED = -2, /* shape outside what the kernels implement */
  TSPN_EWORKSPACE = -3,   /* workspace too small */
  TSPN_ELAUNCH = -4,      /* HIP runtime error at launch */
  TSPN_EDEVICE = -5       /* a kernel of an EARLIER launch raised a fault through the device status block (below) */
};

/* ---- device status block (round 6) ----------------------------------------
 * The reference reports failures as Python exceptions only; a kernel has no such channel, and a
 * synchronisation after every launch to ask it would serialise the host.  Instead the caller allocates
 * TSPN_STATUS_WORDS int32 words of PINNED, DEVICE-MAPPED HOST memory (hipHostMalloc / torch .pin_memory(); zeroed,
 * 64-byte aligned) and attaches them once per device; kernels write them with system-scope atomics, the host reads
 * them with plain loads.  While word TSPN_STATUS_FAULT is non-zero every launch entry of the library on that device
 * returns TSPN_EDEVICE (checked where the HIP launch error is checked: the call that LAUNCHED the faulting kernel
 * returns TSPN_OK, the next one fails) until tspn_status_clear().  Without an attached block a kernel that has to
 * raise traps instead (the launch fails with a HIP error at the next synchronisation).  Attachment is per device
 * (the CURRENT device of the calling thread) and process-wide; the library never frees the block.            */
#define TSPN_STATUS_WORDS 16
enum {
  TSPN_STATUS_FAULT = 0,       /* OR of TSPN_FAULT_* bits; 0 = healthy */
  TSPN_STATUS_FAULT_INFO = 1,  /* raiser's detail (workgroup id of the last raiser) */
  TSPN_STATUS_CONV_ERR = 2,    /* float bits: largest |y - reference| any tspn_conv3_spot_check_f32 has measured since the
                                  host last zeroed the word (not a fault: the host policy decides) */
  TSPN_STATUS_CONV_CHECKS = 3  /* number of outputs spot-checked since the host last zeroed the word */
};
enum {
  TSPN_FAULT_HANDOVER = 1      /* an LDS hand-over between the waves of a workgroup timed out (tspn_bottleneck_tail_io_bf16,
                                  tspn_bottleneck_tail_pipe_bf16): the waiting wave raised and ENDED without touching the
                                  data it did not receive; the launch's output is incomplete */
};
int tspn_status_attach(int32_t* host_words);   /* NULL detaches */
int tspn_status_fault(void);                    /* the fault word of the current device's block (0 if none attached) */
int tspn_status_clear(void);                    /* zero fault + info (the caller has dealt with it) */
/* Raises TSPN_FAULT_HANDOVER on purpose: one wave waits, with a short bound, for an LDS counter that nobody sets --
 * through the same device code as the role-split res4 tail.  `reached_end` (device int32, optional) stays 0: the wave
 * ends inside the wait.  Tests of the channel: the call itself returns TSPN_OK, the next launch entry TSPN_EDEVICE. */
int tspn_status_selftest(int32_t* reached_end, void* stream);

#define TSPN_GEOM_CHANNELS 8

/* tspn_fused_desc.conv_algo: which temporal-conv kernel consumes `conv_packed` (the packing IS the choice) */
enum {
  TSPN_CONV_DIRECT = 0,      /* conv_packed = tspn_pack_conv3_f32(conv.weight, C, C, split = D): [3][D][2C]; any shape */
  TSPN_CONV_WINOGRAD63 = 1   /* conv_packed = tspn_pack_conv3_wino63_frag_f32(conv.weight, C, C, split = D); D % 32 == 0 */
};

int tspn_version(void);
/* sizeof(tspn_fused_desc) / sizeof(tspn_fused_bf16_desc) as this library was compiled: a host whose
 * view of the descriptors differs (stale .so, stale binding) must refuse to call the library.      */
size_t tspn_fused_desc_size(void);
size_t tspn_fused_bf16_desc_size(void);
const char* tspn_last_error(void);
const char* tspn_error_string(int code);

/* ---- a2: predicate head -------------------------------------------------
 * Replaces RelationPredictor.forward (lib/modeling/model.py:85-88):
 *   out[P,K] = sigmoid(x[P,F] @ W[K,F]^T + b[K])       (apply_sigmoid != 0)
 * `ldx` = row stride of x in elements (>= F).  W is the nn.Linear weight as
 * stored in the state_dict (classifier.rel_predictor.weight), no packing.
 * Split-K partial slabs live in `workspace`
 * (tspn_predicate_head_workspace_bytes).                                     */
size_t tspn_predicate_head_workspace_bytes(int64_t P, int64_t F, int64_t K);
int tspn_predicate_head_f32(const float* x, int64_t P, int64_t F, int64_t ldx,
                            const float* W, const float* b, int64_t K,
                            float* out, int apply_sigmoid,
                            void* workspace, size_t workspace_bytes, void* stream);

/* f2 — a15 folded into a2: same as tspn_predicate_head_f32 on RAW features, with the block-L1
 * normalisation of VRDataset._feature_preprocess (lib/dataset/vrdataset.py:219-243) applied on the
 * fly: columns [first + k*block, first + (k+1)*block), k < nblocks, of every row are divided by their
 * L1 norm (0 -> 1).  Split-K slices never straddle a block, each returns sum|x| per row, and the
 * ordered reduce divides — x is read once and never rewritten.                                    */
size_t tspn_predicate_head_norm_workspace_bytes(int64_t P, int64_t F, int64_t K, int64_t first,
                                                int64_t block, int64_t nblocks);
int tspn_predicate_head_norm_f32(const float* x, int64_t P, int64_t F, int64_t ldx,
                                 const float* W, const float* b, int64_t K,
                                 int64_t first, int64_t block, int64_t nblocks,
                                 float* out, int apply_sigmoid,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* ---- a15: block-L1 feature preprocessing --------------------------------
 * Replaces VRDataset._feature_preprocess (lib/dataset/vrdataset.py:219-243,
 * lib/utils/miscellaneous.py:32-35), in place on feats[P, ld>=F].           */
int tspn_feature_preprocess_f32(float* feats, int64_t P, int64_t F, int64_t ld,
                                int64_t first, int64_t block, int64_t nblocks,
                                void* stream);

/* ---- a15: proposal pair filter ------------------------------------------
 * Replaces VRDataset._get_proposal_idx + _get_num_tracklet_proposals
 * (lib/dataset/vrdataset.py:140-148) for S segments stored back to back:
 *   segment s owns pair rows [pair_off[s], pair_off[s+1]) of `pairs` [sum P_s, 2] (track indices LOCAL
 *   to the segment, as in the `pairs` dataset of its -relation.h5 file) and tracks
 *   [track_off[s], track_off[s+1]) of `trackid` (-1 = proposal, >= 0 = ground-truth track).
 *   out_idx[pair_off[s] + r], r < out_count[s]: local indices of the kept pairs (both tracks are
 *       proposals), ascending = the order of the reference's list comprehension;
 *   out_count[s]      number kept, or -1 if a pair of the segment names a track outside it
 *                     (the reference raises IndexError there);
 *   out_num_tracks[s] = sum(trackid < 0).
 * All pointers are device pointers; one workgroup per segment, order-preserving compaction.        */
int tspn_proposal_pair_filter_i64(const int64_t* pairs, const int64_t* pair_off,
                                  const int64_t* trackid, const int64_t* track_off, int64_t S,
                                  int64_t* out_idx, int64_t* out_count, int64_t* out_num_tracks,
                                  void* stream);
/* out[r, 0:F] = src[idx_base + idx[r], 0:F] (src row stride ld): feats[proposal_idx] of
 * vrdataset.py:66-67.                                                                              */
int tspn_gather_rows_f32(const float* src, int64_t ld, int64_t F, const int64_t* idx,
                         int64_t idx_base, int64_t R, float* out, void* stream);

/* ---- a5/a6: PPN pair matrix + top-k pair indices ------------------------
 * Replaces PPNHead.forward + the sort in PPN._forward_test
 * (lib/modeling/relpn/ppn.py:107-112, 84-85) for a batch of `B` segments with
 * N tracklets each:
 *   mat[b] = sigmoid(MLP_s(cls[b]) @ MLP_o(cls[b])^T)        [B,N,N]
 *   idx[b] = first `topk` flat indices (s*N+o) of mat[b] in descending order,
 *            ties broken lower-index-first                   [B,topk] int64
 * MLP = Linear(Cin,H)+ReLU+Linear(H,Cout); weights as in the state_dict
 * (sub_emb.0.weight [H,Cin], sub_emb.2.weight [Cout,H], ...).
 * Limits: N <= 128, Cin,H,Cout <= 256, topk <= N*N.                         */
int tspn_ppn_pair_matrix_topk_f32(const float* cls, int64_t B, int64_t N, int64_t Cin,
                                  int64_t H, int64_t Cout,
                                  const float* ws1, const float* bs1,
                                  const float* ws2, const float* bs2,
                                  const float* wo1, const float* bo1,
                                  const float* wo2, const float* bo2,
                                  int64_t topk, float* out_mat, int64_t* out_idx,
                                  void* stream);

/* ---- a16: trajectory (cubic) IoU ----------------------------------------
 * Replaces cubic_iou/_intersect/_union (lib/modeling/trajectory.py:85-141):
 *   out[B,N1,N2] float32, boxes [B,N,T,4] (l,t,r,b), +1 pixel-inclusive,
 *   intersection accumulated over t in fp32 in frame order.
 * boxes2 == NULL means boxes2 = boxes1 (N2 = N1).                            */
int tspn_traj_iou_f32(const float* boxes1, int64_t N1, const float* boxes2, int64_t N2,
                      int64_t B, int64_t T, float* out, void* stream);

/* ---- f3: association-time trajectory IoU, batched ------------------------
 * Replaces the per-candidate `_traj_iou(r.straj, straj)` / `_traj_iou(r.otraj, otraj)` calls of
 * lib/modeling/association.py:35-48,101-106 (-> trajectory.py:144-158 traj_iou -> cubic_iou on float64
 * boxes) by one launch per segment: a [U,L,4] = every distinct trajectory the previous segment's relations
 * hold, from the current segment's first frame on; len_a[U] int32 = how many of those frames each has (its
 * common frames with the segment's tracklets; 0 = no overlap -> IoU 0, association.py:36-37; the caller
 * guarantees len_a[u] <= L); b [N,L,4] = the current segment's tracklets; out [U,N] float32.  Rounding
 * recipe of the reference chain kept to the bit: coordinates max/min stored as float32, intersection in
 * float32 in frame order, areas float64 in numpy's pairwise summation order, quotient in float64 stored as
 * float32.  The length check is the caller's: this function reads len_a[u] frames of both rows.           */
int tspn_traj_iou_tail_f64(const double* a, const int32_t* len_a, const double* b, int64_t U, int64_t N,
                           int64_t L, float* out, void* stream);

/* ---- pair order ---------------------------------------------------------
 * All ordered pairs (i,j), i != j, i-major (lib/modeling/predict.py:133-140):
 * pairs[N*(N-1), 2] int64, tracklet ids offset by `base`.                   */
int tspn_pair_index_i64(int64_t N, int64_t base, int64_t* pairs, void* stream);

/* ---- N^2 pair builder (materialising form) ------------------------------
 * Builds what DPNHead consumes (rel_feats "NxCxT", lib/modeling/relpn/dpn_anchor.py:38):
 *   out_feat[P, 2D, T] = cat(feats[pairs[p,0]]^T, feats[pairs[p,1]]^T)
 *   out_geom[P, 8, T]  = relative box geometry (may be NULL)
 * feats [NT, T, D], boxes [NT, T, 4] (may be NULL iff out_geom is NULL),
 * pairs int64 [P,2] with ids in [0, NT).  out_feat may be NULL.             */
int tspn_pair_gather_f32(const float* feats, const float* boxes, int64_t NT, int64_t T,
                         int64_t D, const int64_t* pairs, int64_t P,
                         float* out_feat, float* out_geom, void* stream);

/* ---- weight packing for the k=3 temporal conv ---------------------------
 * packed[3][Cin][M] <- W[M][Cin][3] (nn.Conv1d weight layout).  `packed` has
 * M*Cin*3 floats.  With split > 0 the input channels are split at `split`
 * and the two halves stacked along M (factorised pair form):
 *   packed[3][split][2M]: rows [0,M) = W[:, :split, :], rows [M,2M) = W[:, split:, :]
 * (requires Cin == 2*split).                                                */
int tspn_pack_conv3_f32(const float* W, int64_t M, int64_t Cin, int64_t split,
                        float* packed, void* stream);

/* ---- a8: temporal context encoder (k=3 conv as implicit GEMM on MFMA) ----
 * Replaces self.conv + F.relu of DPNHead.forward (lib/modeling/relpn/dpn.py:70):
 *   y[B, M, T] = act(bias[M] + sum_{tap,ci} packed[tap][ci][m] * x[B, ci, t+tap-1])
 * zero padding outside [0,T).  bias may be NULL; relu != 0 applies ReLU.    */
int tspn_conv3_f32(const float* x, int64_t B, int64_t Cin, int64_t T,
                   const float* packed, int64_t M, const float* bias, int relu,
                   float* y, void* stream);

/* Same operator on channels-LAST input x[B, T, Cin] (the tracklet layout [N,T,D] itself), output
 * still y[B, M, T].  Fast path only: needs Cin % 16 == 0, M % 4 == 0, 16-byte aligned x / packed
 * (otherwise TSPN_EUNSUPPORTED: transpose with tspn_transpose_td_f32 and call tspn_conv3_f32).   */
int tspn_conv3_tc_f32(const float* x, int64_t B, int64_t T, int64_t Cin,
                      const float* packed, int64_t M, const float* bias, int relu,
                      float* y, void* stream);

/* Winograd F(6,3) form of tspn_conv3_tc_f32 (tspn_wino63.hip): six output frames from eight inputs, 8 channel-GEMMs
 * on a sixth of the columns = 4/9 of the direct MFMA work (T = 150 tiles exactly: 25 sextets; any T works, a
 * tracklet's last sextet is masked).  The input transform V = B^T d (points 0, +-1, +-2, +-1/2, inf) is its
 * own HBM-bound pass into `workspace` (tspn_conv3_tc_wino63_workspace_bytes), the MFMA kernel stages it by LDS-DMA.
 *   frag = tspn_pack_conv3_wino63_frag_f32(conv.weight [M, Cin, 3], split): U_j = G g computed in double, rounded
 *          once, stored fragment-major [M'/32][Cin'/8][8][64 lanes][4] (split as in tspn_pack_conv3_f32)
 * fp32 error against float64, measured at K = 3 x 2048 (tests/test_gpu_wino63.py, profiles/r3/conv_error_realistic.txt):
 * |err| <= 64 eps sum_k |x_k||w_k| per output element (direct form: <= 16 eps ...); on temporally smooth
 * features it equals the direct form's error, on temporally independent heavy-tailed ones it is up to ~5x larger
 * (1.3e-5 of max|y|).  Callers that need the direct form's error pass TSPN_CONV_DIRECT (2.25x the MFMA work).
 * Needs Cin % 32 == 0, M % 32 == 0, 16-byte aligned operands.  tspn_forward_fused_f32 runs it for
 * conv_algo TSPN_CONV_WINOGRAD63.                                                                           */
int tspn_pack_conv3_wino63_frag_f32(const float* W, int64_t M, int64_t Cin, int64_t split, float* frag,
                                    void* stream);
size_t tspn_conv3_tc_wino63_workspace_bytes(int64_t B, int64_t T, int64_t Cin);
int tspn_conv3_tc_wino63_f32(const float* x, int64_t B, int64_t T, int64_t Cin, const float* frag,
                             int64_t M, const float* bias, int relu, float* y,
                             void* workspace, size_t workspace_bytes, void* stream);
/* How the MFMA kernel addresses the transformed input: 0 (default) = buffer loads with 32-bit offsets where the
 * workspace is below 4 GB, 64-bit pointers above; 1 = pointers everywhere.  Both forms land the same bytes in LDS
 * (tests compare them bit for bit).  Process-wide; returns the previous value, TSPN_EINVAL for any other `form`. */
int tspn_conv3_tc_wino63_set_piece_form(int form);



/* ---- a8/a10: relationness + span-regression heads -----------------------
 * Replaces duration_pred (lib/modeling/relpn/dpn.py:71) and relness_pred
 * (lib/modeling/relpn/dpn_anchor.py:105) as ONE [H, C] 1x1 GEMM:
 *   out[P, H, T] = bh[H] + Wh[H, C] @ h_p[C, T]
 * with h_p formed on the fly (never stored):
 *   mode 0 (dense):      h_p = a[ia[p]]                         (a = ReLU'd encoder output [*,C,T])
 *   mode 1 (factorised): h_p = relu(a[ia[p]] + b[ib[p]] + bias) (a,b = tracklet projections)
 * a, b are [*, lda, T] tensors whose first C channels are used
 * (row stride lda*T floats); ia/ib are int64 row ids read at ia[p*idx_stride]
 * (NULL = identity; idx_stride = 2 walks one column of a [P,2] pair table).
 * H <= 16.                                                                  */
int tspn_heads_f32(int mode, const float* a, const float* b, int64_t lda,
                   const int64_t* ia, const int64_t* ib, int64_t idx_stride, const float* bias,
                   const float* Wh, const float* bh, int64_t H,
                   int64_t P, int64_t C, int64_t T, float* out, void* stream);

/* Blocked form of mode 1 for the CANONICAL pair table (all ordered pairs (s,o), s != o, s-major,
 * per video; lib/modeling/predict.py:133-140): y[B*N, 2C, T] holds the tracklet projections
 * (channels [0,C) subject part incl. bias, [C,2C) object part);
 *   out[b*N*(N-1) + s*(N-1) + o - (o>s)][H][T] = bh + Wh @ relu(y[b*N+s, :C] + y[b*N+o, C:])
 * 8x8 pair blocks share their 16 rows through LDS (6x less traffic than tspn_heads_f32).       */
int tspn_heads_pairgrid_f32(const float* y, int64_t B, int64_t N, int64_t C, int64_t T,
                            const float* Wh, const float* bh, int64_t H, float* out,
                            void* stream);

/* ---- a3: RelOIPool over time --------------------------------------------
 * mean over t of x[R, T, D] -> out[R, D]   (layout_tc = 1, tracklet layout)
 * mean over t of x[R, C, T] -> out[R, C]   (layout_tc = 0, channels-first)  */
int tspn_temporal_mean_f32(const float* x, int64_t R, int64_t T, int64_t Cdim,
                           int layout_tc, float* out, void* stream);
/* sum over t of x[R, T, D] -> out[R, D], frames added in order (with R = 1: the column sums of a matrix -- the bias
 * gradients of the training step, lib/modeling/train.py:74-78; round 4)                                          */
int tspn_temporal_sum_f32(const float* x, int64_t R, int64_t T, int64_t Cdim, float* out, void* stream);

/* gather rows: out[P, 2D] = cat(src[pairs[p,0]], src[pairs[p,1]])           */
int tspn_pair_rows_f32(const float* src, int64_t NT, int64_t D, const int64_t* pairs,
                       int64_t P, float* out, void* stream);

/* [R,T,D] -> [R,D,T] (tracklet layout -> channels-first)                     */
int tspn_transpose_td_f32(const float* x, int64_t R, int64_t T, int64_t D, float* out,
                          void* stream);

/* ---- f1: top-k triplet decode -------------------------------------------
 * Replaces the per-segment decode of lib/modeling/predict.py:66-106 for S segments of P pairs:
 * per pair the topk_pair best of the K predicate scores, per segment the topk_seg best of those
 * (both "larger first, lower index first on ties"), then for winner r of segment s:
 *   out_score[s,r]        the score
 *   out_pair_tid[s,r,:]   = pairs[s, pair(r), :]                    (local tracklet ids)
 *   out_triplet[s,r,:]    = (argmax cls_sub[row(tid_s)], predicate id, argmax cls_obj[row(tid_o)])
 * with row(tid) = s*seg_rows + row_mul*tid in a matrix of row stride `ld`, NO class columns.
 * predict.py:88-89 reads row (N-1)*tid of the pair-feature matrix (columns 0:35 / 35:70): pass
 * cls_sub = feats, cls_obj = feats + 35, ld = F, seg_rows = P, row_mul = N-1 to reproduce it, or
 * the per-tracklet class logits with seg_rows = N, row_mul = 1.
 * M = min(topk_seg, P*min(topk_pair,K)) <= 1024 rows are written per segment; K <= 256.          */
size_t tspn_decode_topk_workspace_bytes(int64_t S, int64_t P, int64_t topk_pair);
int tspn_decode_topk_f32(const float* rel_logit, const int64_t* pairs,
                         const float* cls_sub, const float* cls_obj, int64_t ld,
                         int64_t seg_rows, int64_t row_mul, int64_t S, int64_t P, int64_t K,
                         int64_t NO, int64_t topk_pair, int64_t topk_seg,
                         float* out_score, int64_t* out_triplet, int64_t* out_pair_tid,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ---- f3: temporal span decode + 1-D NMS ---------------------------------
 * Build-defined completion of the reference's stub RelNMS (lib/modeling/relpn/rel_nms.py:5-15;
 * anchors per lib/modeling/relpn/anchor_generator.py:48-59) — semantics in DESIGN.md §2 and
 * oracle.decode_spans.  heads[P, 3A, T] as produced by tspn_forward_fused_f32 /
 * tspn_temporal_encoder_heads_f32 (rows [0,A) relationness logits, [A,3A) (d_c, d_w) per anchor);
 * `sizes_host` = A anchor widths in frames (HOST pointer, copied into the launch).
 * Per pair the first `top_k` NMS survivors, best first:
 *   out_anchor[P,top_k] candidate index t*A+a (-1 = unused)   out_span[P,top_k,2] int64 frames [s,e)
 *   out_span_f[P,top_k,2] fp32 (start,end)                    out_score[P,top_k] sigmoid(logit)
 *   out_count[P] number of survivors.
 * Limits: A <= 8, A*T <= 4096, top_k <= 1024; at most min(pre_nms, 1024) candidates enter the NMS. */
int tspn_decode_spans_f32(const float* heads, int64_t P, int64_t A, int64_t T,
                          const float* sizes_host, int64_t top_k, double nms_threshold,
                          int64_t pre_nms, int64_t* out_anchor, int64_t* out_span,
                          float* out_span_f, float* out_score, int64_t* out_count, void* stream);

/* ---- whole relation-scoring pass on tracklet tensors --------------------
 * The fused/factorised product path for `B` videos of N
 * tracklets each: pair builder + temporal encoder + relationness/span heads
 * + RelOIPool + predicate head, without materialising [P, 2D, T].           */
typedef struct tspn_fused_desc {
  int64_t B, N, T, D;          /* videos, tracklets per video, frames, RoI dim; C = 2D */
  int64_t A, K;                /* anchors per location; predicates */
  const float* feats;          /* [B*N, T, D] */
  const int64_t* pairs;        /* [P,2] global tracklet ids (video b: b*N + local) */
  int64_t P;
  int64_t canonical_pairs;     /* != 0: `pairs` is the canonical table of tspn_pair_index_i64 for every
                                  video in order (P == B*N*(N-1)): enables the blocked pair stage */
  const float* conv_packed;    /* packed conv.weight [C,C,3], see TSPN_CONV_* above */
  int64_t conv_algo;           /* TSPN_CONV_DIRECT (k=3 taps as an implicit GEMM, any shape) or
                                  TSPN_CONV_WINOGRAD63 (Winograd F(6,3), 4/9 of the MFMA work, D % 32 == 0) */
  const float* conv_bias;      /* [C] */
  const float* head_w;         /* [3A, C]: rows [0,A) relness_pred, [A,3A) duration_pred */
  const float* head_b;         /* [3A] */
  const float* cls_w;          /* [K, C] */
  const float* cls_b;          /* [K] */
  float* out_heads;            /* [P, 3A, T] */
  float* out_logits;           /* [P, K] */
  void* workspace;
  size_t workspace_bytes;
  /* optional profiling hooks: hipEvent_t recorded on `stream` immediately before / after the
   * dominant kernel (the tracklet-projection implicit GEMM); NULL = off */
  void* ev_conv_begin;
  void* ev_conv_end;
  /* optional: hipEvent_t recorded on `stream` as soon as out_logits is complete.  The predicate logits depend
   * only on the tracklet means, so the driver computes them FIRST; a caller can start the top-k decode, the PPN
   * and the result gather on a second stream behind this event while the encoder is still running. */
  void* ev_logits_ready;
  /* optional accuracy guard of TSPN_CONV_WINOGRAD63 (round 6; tspn_conv3_spot_check_f32 below): the RAW conv.weight
   * [C, C, 3] and the number of output rows to spot-check per call (0 / NULL = off; needs an attached status block) */
  const float* conv_weight;
  int64_t conv_check;
} tspn_fused_desc;

size_t tspn_forward_fused_workspace_bytes(const tspn_fused_desc* d);
int tspn_forward_fused_f32(const tspn_fused_desc* d, void* stream);

/* ---- a-posteriori accuracy guard of the temporal conv (round 6) ------------
 * The encoder's contract is a plain fp32 Conv1d (lib/modeling/relpn/dpn.py:69-73).  Behind a conv launch, `rows`
 * workgroups each pick one output row (stratified over the rows, a different draw every call) and recompute that
 * row at up to 24 columns -- the six frames of four 6-frame groups ("sextets": the one named in scratch word 0, three hashed
 * ones) -- in float64 from the RAW weights, and compare with y:
 *   x [B, T, Cin]; W = conv.weight [M, Cw, 3] with `split` as in tspn_pack_conv3_f32 (split > 0: Cw = 2 split,
 *   Cin = split, y has 2M rows: [0, M) from W[:, :split], [M, 2M) from W[:, split:]); bias per y row or NULL;
 *   y [B, rows of y, ldy] (ldy >= T);  scratch: TSPN_CONV_CHECK_SCRATCH_BYTES of 8-byte aligned DEVICE memory, zeroed
 *   by the caller before the conv it belongs to: bytes [0, 32) = where the workgroups meet (the kernel leaves them
 *   zeroed); from TSPN_CONV_CHECK_HOT_OFFSET on, TSPN_CONV_CHECK_HOT_SLOTS 64-bit keys 256 bytes apart, into which the
 *   F(6,3) input transform reports (float bits of the largest |x| a wave saw) << 32 | its sextet b * ceil(T/6) + q:
 *   the sextet of the largest key is the first of the four checked -- the Winograd error peaks in the sextet that holds
 *   an input outlier.  All keys 0 = four hashed sextets.
 * Result (no fault): status word TSPN_STATUS_CONV_ERR = max(itself, float bits of the largest |y - y_ref|),
 * TSPN_STATUS_CONV_CHECKS += outputs checked -- written once per launch by the last workgroup to finish.  Needs an
 * attached status block.                                                                                          */
#define TSPN_CONV_CHECK_HOT_SLOTS 64
#define TSPN_CONV_CHECK_HOT_OFFSET 256
#define TSPN_CONV_CHECK_SCRATCH_BYTES (TSPN_CONV_CHECK_HOT_OFFSET + 256 * TSPN_CONV_CHECK_HOT_SLOTS)
int tspn_conv3_spot_check_f32(const float* x, int64_t B, int64_t T, int64_t Cin, const float* W, int64_t M,
                              int64_t Cw, int64_t split, const float* bias, int relu, const float* y, int64_t ldy,
                              uint64_t* scratch, int64_t rows, void* stream);

/* ---- span-restricted RelOIPool + predicate head --------------------------
 * Build-defined meaning of RelOIPool with duration proposals (reference lib/modeling/model.py:68-73 indexes
 * a list with a tensor and cannot run): the pair feature cat(f_s, f_o) is averaged over
 * the pair's own span [start, end) of frames before RelationPredictor (model.py:85-88):
 *   out[p] = sigmoid(cls_w . mean_{t in [start_p, end_p)} cat(f[s_p, t], f[o_p, t]) + cls_b).
 * feats [NT, T, D]; pairs [P,2] global tracklet ids; spans int64 [P,2], e.g. the top span of
 * tspn_decode_spans_f32 (a negative start selects the whole segment).                               */
size_t tspn_span_predicate_workspace_bytes(int64_t NT, int64_t T, int64_t D, int64_t K);
int tspn_span_predicate_f32(const float* feats, int64_t NT, int64_t T, int64_t D, const int64_t* pairs,
                            const int64_t* spans, int64_t P, const float* cls_w, const float* cls_b,
                            int64_t K, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- bf16-operand path (BASELINE config 3: N=64, T=900, D=1024, bf16) ----
 * Semantics (build-defined; pinned by tests/golden/g8 against the reference's own DPNHead /
 * RelationPredictor modules cast with .bfloat16(), lib/modeling/relpn/dpn.py:55-73, model.py:76-88):
 * operands bf16 (stored as uint16_t bit patterns), products exact, accumulation and biases fp32; the
 * encoder activation relu(conv + b) is rounded to bf16 once, the span-pooled feature is rounded to
 * bf16, head outputs and logits are fp32.                                                          */
int tspn_cast_bf16(const float* src, int64_t n, uint16_t* dst, void* stream); /* round-to-nearest-even */
/* conv.weight [M, Cin, 3] fp32 -> [3][Cp/8][Mp][8] bf16; `split` as in tspn_pack_conv3_f32 */
int tspn_pack_conv3_bf16(const float* W, int64_t M, int64_t Cin, int64_t split, uint16_t* packed,
                         void* stream);
/* 1x1 head weights [H <= 16, C] fp32 -> [C/8][16][8] bf16 (rows H..15 zero) */
int tspn_pack_heads_bf16(const float* W, int64_t H, int64_t C, uint16_t* packed, void* stream);
/* k=3, pad=1 conv over time; x bf16 channels-last [B, T, Cin]; y fp32 channels-last [B*T, ldm]
 * (y[n][m], n = b*T + t), + bias[m] if given.  Needs Cin % 16 == 0, M % 4 == 0, ldm % 4 == 0 and packed weights below
 * 2 GB (the operand pieces are buffer loads with 32-bit offsets; x may have any size). */
int tspn_conv3_tc_bf16(const uint16_t* x, int64_t B, int64_t T, int64_t Cin, const uint16_t* packed,
                       int64_t M, const float* bias, float* y, int64_t ldm, void* stream);
/* pair stage on the canonical pair table: y fp32 [B*N*T, ldm] with U = channels [0,C), V = [C,2C) (ONE video's rows,
 * N*T*ldm*4 bytes, must stay below 2 GB: buffer loads with 32-bit offsets from the video's base);
 * out[p][h][t] = head_b[h] + sum_c Wh[h][c] * bf16(relu(U[s][t][c] + V[o][t][c])), out [B*N*(N-1), H, T] */
int tspn_heads_pairgrid_bf16(const float* y, int64_t ldm, int64_t B, int64_t N, int64_t C, int64_t T,
                             const uint16_t* head_packed, const float* head_b, int64_t H, float* out,
                             void* stream);
/* mean over frames of bf16 [R, T, D] -> fp32 [R, D] holding bf16-rounded values (RelOIPool over the
 * whole segment, model.py:59-60) */
int tspn_temporal_mean_bf16(const uint16_t* x, int64_t R, int64_t T, int64_t D, float* out, void* stream);

typedef struct tspn_fused_bf16_desc {
  int64_t B, N, T, D;            /* C = 2D; D % 16 == 0 */
  int64_t A, K;
  const uint16_t* feats;         /* bf16 [B*N, T, D] */
  const int64_t* pairs;          /* canonical table of tspn_pair_index_i64 for every video, global ids */
  int64_t P;                     /* == B*N*(N-1) */
  const uint16_t* conv_packed;   /* tspn_pack_conv3_bf16(conv.weight [C,C,3], split=D): [3][D/8][2C][8] */
  const float* conv_bias;        /* [C] fp32 */
  const uint16_t* head_packed;   /* tspn_pack_heads_bf16([3A, C]) */
  const float* head_b;           /* [3A] fp32 */
  const float* cls_w;            /* [K, C] fp32 storage of bf16-rounded weights (products of bf16 values are
                                    exact in fp32, so the fp32 predicate kernels give the bf16-MFMA result) */
  const float* cls_b;            /* [K] */
  float* out_heads;              /* [P, 3A, T] fp32 */
  float* out_logits;             /* [P, K] fp32 */
  void* workspace;
  size_t workspace_bytes;
  void* ev_conv_begin;           /* optional hipEvent_t around the conv kernel, as in tspn_fused_desc */
  void* ev_conv_end;
  void* ev_logits_ready;         /* optional, as in tspn_fused_desc: recorded as soon as out_logits is complete (the logits
                                    are computed first) */
} tspn_fused_bf16_desc;

size_t tspn_forward_fused_bf16_workspace_bytes(const tspn_fused_bf16_desc* d);
int tspn_forward_fused_bf16(const tspn_fused_bf16_desc* d, void* stream);

/* ---- dense reference-faithful encoder + heads on a materialised [P,C,T] --
 * DPNHead.forward (lib/modeling/relpn/dpn.py:69-73) on arbitrary pair feats:
 * conv3+ReLU into `h_ws` [P,C,T] (caller scratch), then tspn_heads_f32 mode 0. */
int tspn_temporal_encoder_heads_f32(const float* x, int64_t P, int64_t C, int64_t T,
                                    const float* conv_packed, const float* conv_bias,
                                    const float* head_w, const float* head_b, int64_t H,
                                    float* h_ws, float* out_heads, void* stream);

/* The same on bf16 operands (semantics of tspn_forward_fused_bf16: bf16 x / weights, exact products, fp32
 * accumulation, relu(conv + b) rounded to bf16 once, fp32 outputs) -- what `DPNHead.bfloat16()` computes
 * (lib/modeling/relpn/dpn.py:55-73), pinned by golden g8.
 *   tspn_transpose_cast_bf16   x [P,C,T] fp32 (DPNHead's input layout) -> bf16 channels-last [P,T,C]
 *   tspn_heads_dense_bf16      out[p][h][t] = head_b[h] + sum_c Wh[h][c] * bf16(relu(y[p][t][c])); y fp32 [P,T,ldm]
 *   tspn_temporal_encoder_heads_bf16 = tspn_conv3_tc_bf16 (conv_packed = tspn_pack_conv3_bf16(conv.weight, split=0),
 *                                conv_bias fp32 holding bf16 values or NULL) into y_ws [P,T,C] fp32, then the heads
 *                                (head_packed = tspn_pack_heads_bf16).  Needs C % 32 == 0, H <= 16.                 */
int tspn_transpose_cast_bf16(const float* x, int64_t P, int64_t C, int64_t T, uint16_t* out, void* stream);
int tspn_heads_dense_bf16(const float* y, int64_t ldm, int64_t P, int64_t C, int64_t T,
                          const uint16_t* head_packed, const float* head_b, int64_t H, float* out, void* stream);
int tspn_temporal_encoder_heads_bf16(const uint16_t* x_tc, int64_t P, int64_t C, int64_t T,
                                     const uint16_t* conv_packed, const float* conv_bias,
                                     const uint16_t* head_packed, const float* head_b, int64_t H,
                                     float* y_ws, float* out_heads, void* stream);

/* ---- f4 (first slice): RoI feature head ------------------------------------------------------
 * The reference extracts tracklet RoI features with detectron2's R101-C4 model, configured in
 * lib/detectron/trainer.py:23-33 (no code of its own): ROIAlign 14x14 on the res4 map -> res5
 * (3 bottleneck blocks, FrozenBN) -> mean over 7x7.  These are the operators of that head on
 * channels-last (NHWC) fp32 tensors; `temporal-span-proposal-network-vidvrd_amd/roi_head.py`
 * assembles them and hands [N,T,2048] straight to the pair builder.
 *
 * tspn_pack_conv2d_f32: Conv2d weight [Cout][Cin][KH][KW] -> [KH*KW][Cin][Cout] (KH*KW <= 64).
 * tspn_conv2d_nhwc_f32: out[NB,OH,OW,Cout] = act( conv(x[NB,H,W,Cin]) + bias[Cout] + residual[NB,OH,OW,Cout] ),
 *   zero padding `pad`, `stride`, OH = (H + 2 pad - KH) / stride + 1; bias / residual may be NULL;
 *   relu != 0 applies max(.,0).  BatchNorm is folded into (packed, bias) by the caller.  Implicit GEMM
 *   on fp32 MFMA.  Needs Cin % 16 == 0, Cout % 4 == 0, 16-byte aligned tensors (else TSPN_EUNSUPPORTED). */
int tspn_pack_conv2d_f32(const float* w, int64_t Cout, int64_t Cin, int64_t KH, int64_t KW,
                         float* packed, void* stream);
int tspn_conv2d_nhwc_f32(const float* x, int64_t NB, int64_t H, int64_t W, int64_t Cin,
                         const float* packed, int64_t Cout, int64_t KH, int64_t KW, int64_t stride,
                         int64_t pad, const float* bias, const float* residual, int relu, float* out,
                         void* stream);
/* Fast variant for Cout % 32 == 0 (the res5 widths): FRAGMENT-MAJOR weights
 *   frag[Cout/32][KH*KW][Cin/16][64 lanes = 32 kh + li][8 = (g, r)] = w[32 mb + li][16 c + 4 g + 2 kh + r][tap]
 * loaded straight into MFMA operand registers (each wave owns 32 output rows; only x goes through LDS).
 * Same arguments and result as tspn_conv2d_nhwc_f32 (bit-identical: same contraction order). */
int tspn_pack_conv2d_frag_f32(const float* w, int64_t Cout, int64_t Cin, int64_t KH, int64_t KW,
                              float* frag, void* stream);
int tspn_conv2d_nhwc_frag_f32(const float* x, int64_t NB, int64_t H, int64_t W, int64_t Cin,
                              const float* frag, int64_t Cout, int64_t KH, int64_t KW, int64_t stride,
                              int64_t pad, const float* bias, const float* residual, int relu, float* out,
                              void* stream);
/* Stem form for Cin <= 4 (RGB): x[NB,H,W,4] (channels zero-padded to 4); one K chunk = four taps x 4 channels,
 * so a 7x7 stem runs 13 chunks instead of 49 on 16-channel padding.
 *   frag[Cout/32][ceil(KH*KW/4)][64 lanes][8 = (g, r)] = w[32 mb + li][2 kh + r][tap = 4 c + g] (0 beyond Cin / taps) */
int tspn_pack_conv2d_frag_cin4_f32(const float* w, int64_t Cout, int64_t Cin, int64_t KH, int64_t KW,
                                   float* frag, void* stream);
int tspn_conv2d_nhwc_cin4_f32(const float* x, int64_t NB, int64_t H, int64_t W, const float* frag,
                              int64_t Cout, int64_t KH, int64_t KW, int64_t stride, int64_t pad,
                              const float* bias, int relu, float* out, void* stream);
/* bf16-operand form (tspn_roi_bf16.hip; v_mfma_f32_32x32x16_bf16): x, residual, out are bf16 (uint16_t
 * bit patterns), bias fp32; products exact, fp32 accumulation, act(acc + bias + residual) rounded to bf16
 * once (round to nearest even).  Weights: tspn_pack_conv2d_frag_bf16 rounds the fp32 (BN-folded) weight
 * once into  frag[Cout/32][Cin/64][KH*KW][4 ks][64 lanes = 32 kh + li][8 j] =
 * bf16(w[32 mb + li][64 c + 16 ks + 8 kh + j][tap])  (ABI 5: channel chunk outermost -- the contraction runs chunk by
 * chunk, all taps of a chunk in a row, in tspn_conv2d_nhwc_bf16 and in tspn_bottleneck_tail_bf16 alike).
 * Needs Cin % 64 == 0, Cout % 32 == 0. */
int tspn_pack_conv2d_frag_bf16(const float* w, int64_t Cout, int64_t Cin, int64_t KH, int64_t KW,
                               uint16_t* frag, void* stream);
int tspn_conv2d_nhwc_bf16(const uint16_t* x, int64_t NB, int64_t H, int64_t W, int64_t Cin,
                          const uint16_t* frag, int64_t Cout, int64_t KH, int64_t KW, int64_t stride,
                          int64_t pad, const float* bias, const uint16_t* residual, int relu,
                          uint16_t* out, void* stream);
/* tspn_roi_align_nhwc_f32: detectron2 ROIAlign on a channels-last map feat[NF,H,W,C]:
 *   rois[R,5] = (map index, x1, y1, x2, y2) in image coordinates, `spatial_scale` image -> map,
 *   out[R,OP,OP,C]; sampling_ratio 0 = adaptive grid ceil(roi size / P) (detectron2's POOLER_SAMPLING_RATIO 0);
 *   aligned != 0 = the half-pixel-corrected form (ROIAlignV2).  `bin_stride` >= 1: only the bins (bs i, bs j) of the
 *   P x P grid are produced, OP = ceil(P / bs) -- with bs = 2 exactly the bins the stride-2 1x1 convolutions of res5's
 *   first block read (stride_in_1x1), a quarter of the work and of the output.  Needs C % 4 == 0. */
int tspn_roi_align_nhwc_f32(const float* feat, int64_t NF, int64_t H, int64_t W, int64_t C,
                            const float* rois, int64_t R, int64_t P, float spatial_scale,
                            int sampling_ratio, int aligned, int bin_stride, float* out, void* stream);
/* bf16 map in (values exact in fp32, interpolation in fp32), bf16 out */
int tspn_roi_align_nhwc_bf16(const uint16_t* feat, int64_t NF, int64_t H, int64_t W, int64_t C,
                             const float* rois, int64_t R, int64_t P, float spatial_scale,
                             int sampling_ratio, int aligned, int bin_stride, uint16_t* out, void* stream);
/* same interpolation in fp32, result rounded once to bf16 (input of tspn_conv2d_nhwc_bf16) */
int tspn_roi_align_nhwc_f32_bf16out(const float* feat, int64_t NF, int64_t H, int64_t W, int64_t C,
                                    const float* rois, int64_t R, int64_t P, float spatial_scale,
                                    int sampling_ratio, int aligned, int bin_stride, uint16_t* out, void* stream);

/* max_pool2d(k, stride, pad) on a channels-last fp32 map x[NB,H,W,C] -> out[NB,OH,OW,C], fp32 (out_bf16 == 0)
 * or bf16 (rounded once); padding positions do not take part.  detectron2 BasicStem uses 3 / 2 / 1.
 * Needs C % 4 == 0. */
int tspn_max_pool_nhwc_f32(const float* x, int64_t NB, int64_t H, int64_t W, int64_t C, int64_t k,
                           int64_t stride, int64_t pad, void* out, int out_bf16, void* stream);

/* ---- f4: fused tail of a bottleneck block on bf16 operands (tspn_bottleneck_bf16.hip) -----------------------
 * detectron2 BottleneckBlock.forward after conv1 (modeling/backbone/resnet.py), FrozenBN folded by the caller:
 *     out = relu( W3 . relu(W2 (*) h1 + b2) + b3 + residual )      conv2 = 3x3 / pad 1 / stride 1, conv3 = 1x1
 * h1 bf16 [NB,H,W,CM] (conv1's output), residual / out bf16 [NB,H,W,4 CM] (the block input or its projection
 * shortcut); CM = 64, 128 or 256.  frag2 = tspn_pack_conv2d_frag_bf16(W2 [CM,CM,3,3]), frag3 =
 * tspn_pack_conv2d_frag_bf16(W3 [4 CM,CM,1,1]); biases fp32.  Same rounding points and contraction order as
 * tspn_conv2d_nhwc_bf16 applied twice (h2 rounded to bf16 once): bit-identical results, one launch, h2 never in HBM. */
int tspn_bottleneck_tail_bf16(const uint16_t* h1, int64_t NB, int64_t H, int64_t W, int64_t CM,
                              const uint16_t* frag2, const float* bias2, const uint16_t* frag3,
                              const float* bias3, const uint16_t* residual, uint16_t* out, void* stream);
/* The same operator at CM = 256 with the workgroup's waves split by ROLE (tspn_tail_io_bf16.hip, round 5): four waves issue
 * every MFMA and read only weights from L2, four waves do everything that touches HBM (the LDS-DMA of the h1 ranges, the
 * residual rows, the epilogue and the stores); fp32 sums change hands through LDS under two counters per wave pair (no
 * workgroup barrier in the expand phase), the epilogue's global accesses are line-major.  Bit-identical to
 * tspn_bottleneck_tail_bf16 and 16 - 20 % faster at the res4 shape (profiles/r5/tail_role_split.md); what
 * roi_head.BottleneckBlock launches for 256 bottleneck channels.  One workgroup of 512 threads and 147 KB of LDS per CU. */
int tspn_bottleneck_tail_io_bf16(const uint16_t* h1, int64_t NB, int64_t H, int64_t W, int64_t CM,
                                 const uint16_t* frag2, const float* bias2, const uint16_t* frag3,
                                 const float* bias3, const uint16_t* residual, uint16_t* out, void* stream);

/* A whole identity-shortcut bottleneck block in ONE launch (tspn_block_bf16.hip; detectron2 BottleneckBlock.forward,
 * modeling/backbone/resnet.py, with stride 1 and no projection shortcut -- every block of a stage but its first):
 *   out = relu(W3 . relu(W2 (*) relu(W1 . x + b1) + b2) + b3 + x)
 * x, out bf16 channels-last [NB, H, W, 4 CM]; frag1 / frag2 / frag3 = tspn_pack_conv2d_frag_bf16 of the folded
 * 1x1 (4 CM -> CM), 3x3 (CM -> CM) and 1x1 (CM -> 4 CM) weights; fp32 biases.  CM = 64 or 128 (the memory-bound stages
 * res2 / res3): the 4 CM-channel map is read once and h1 / h2 stay on the CU.  Bit-identical to tspn_conv2d_nhwc_bf16
 * (conv1) followed by tspn_bottleneck_tail_bf16.  out must not alias x; one image's map below 2 GB. */
int tspn_bottleneck_block_bf16(const uint16_t* x, int64_t NB, int64_t H, int64_t W, int64_t CM,
                               const uint16_t* frag1, const float* bias1, const uint16_t* frag2, const float* bias2,
                               const uint16_t* frag3, const float* bias3, uint16_t* out, void* stream);
/* The FIRST block of a stage in one launch: its 1x1 convs (conv1 and the projection shortcut) read the input x
 * [NB, Hin, Win, CIN] with stride `stride`, and
 *   out = relu(W3 . relu(W2 (*) relu(W1 . x_s + b1) + b2) + b3 + bf16(Ws . x_s + bs)),   x_s = every stride-th pixel of x,
 * out [NB, (Hin-1)/stride+1, (Win-1)/stride+1, 4 CM]: the shortcut map (4 CM channels) is neither written nor read back.
 * frags = tspn_pack_conv2d_frag_bf16 of the folded shortcut weights [4 CM, CIN, 1, 1].  Built for detectron2's res2.0
 * (CIN = 64, CM = 64, stride 1).  Bit-identical to the four launches it
 * replaces (conv1, shortcut, fused tail). */
int tspn_bottleneck_block_proj_bf16(const uint16_t* x, int64_t NB, int64_t Hin, int64_t Win, int64_t CIN, int64_t stride,
                                    int64_t CM, const uint16_t* frag1, const float* bias1, const uint16_t* frag2,
                                    const float* bias2, const uint16_t* frag3, const float* bias3, const uint16_t* frags,
                                    const float* biass, uint16_t* out, void* stream);
/* The first block of res3 (CIN = 256, CM = 128, stride 2): conv1 (1x1, stride 2) + 3x3 + expand + residual + ReLU in one
 * launch, the residual [NB, H, W, 4 CM] handed over (the output of the separately launched projection shortcut):
 *   out = relu(W3 . relu(W2 (*) relu(W1 . x_s + b1) + b2) + b3 + residual).
 * Bit-identical to tspn_conv2d_nhwc_bf16 (conv1) + tspn_bottleneck_tail_bf16. */
int tspn_bottleneck_block_res_bf16(const uint16_t* x, int64_t NB, int64_t Hin, int64_t Win, int64_t CIN, int64_t stride,
                                   int64_t CM, const uint16_t* frag1, const float* bias1, const uint16_t* frag2,
                                   const float* bias2, const uint16_t* frag3, const float* bias3, const uint16_t* residual,
                                   uint16_t* out, void* stream);

/* The same launch + conv1 of the FOLLOWING block on the tile it has just produced (round 4; CM = 256 = every res4
 * block of an R-50 / R-101 C4 backbone): additionally
 *     h1n = relu( W1n . out + b1n )        1x1, 4 CM -> CM channels, bf16 [NB,H,W,CM]
 * with frag1n = tspn_pack_conv2d_frag_bf16(W1n [CM,4 CM,1,1]) of the next block's conv1 (stride 1).  The 4 CM-channel
 * map is then read once per block (as the residual) instead of twice.  out and h1n are bit-identical to
 * tspn_bottleneck_tail_bf16 followed by tspn_conv2d_nhwc_bf16(out, frag1n, relu) (same contraction order: channels
 * 0..4 CM - 1 in k-steps of 16 on one accumulator chain). */
int tspn_bottleneck_tail_next_bf16(const uint16_t* h1, int64_t NB, int64_t H, int64_t W, int64_t CM,
                                   const uint16_t* frag2, const float* bias2, const uint16_t* frag3,
                                   const float* bias3, const uint16_t* residual, uint16_t* out,
                                   const uint16_t* frag1n, const float* bias1n, uint16_t* h1n, void* stream);

/* tspn_bottleneck_tail_bf16 as a PERSISTENT kernel pipelined across tiles (round 4, tspn_bottleneck_pipe_bf16.hip; CM =
 * 256): one workgroup of eight waves per CU walks the 128-pixel tiles; four waves run the 3x3 phase of tile t while the
 * other four run the expand + residual + store of tile t - 1 (x ring and one h2 image side by side in 136 KB of LDS, 14
 * workgroup barriers per tile on both sides).  Same arithmetic, bit-identical results.  max_workgroups: 0 = one per CU
 * (what it is built for); a smaller positive number makes every workgroup walk more tiles (tests). */
int tspn_bottleneck_tail_pipe_bf16(const uint16_t* h1, int64_t NB, int64_t H, int64_t W, int64_t CM,
                                   const uint16_t* frag2, const float* bias2, const uint16_t* frag3,
                                   const float* bias3, const uint16_t* residual, uint16_t* out,
                                   int64_t max_workgroups, void* stream);

/* ---- f4: bf16-operand stem of the C4 backbone (tspn_stem_bf16.hip) ----------------------------------------
 * detectron2 BasicStem conv (modeling/backbone/resnet.py: 7x7, stride 2, padding 3, RGB in, FrozenBN folded by the
 * caller into w / bias) + ReLU with bf16 operands: image and weights rounded to bf16, exact products, fp32
 * accumulation, relu(acc + bias) rounded to bf16 once.  x fp32 [NB,H,W,3] -> out bf16 [NB,OH,OW,Cout],
 * OH = (H - 1) / 2 + 1.  Cout = 32 or 64.  `frag` = tspn_pack_stem_bf16(w fp32 [Cout,3,7,7]) (Cout * 256 bf16);
 * `workspace` (tspn_stem_bf16_workspace_bytes) holds the 2x2 space-to-depth image the conv kernel streams.
 * tspn_max_pool_nhwc_bf16: max_pool2d(k, stride, pad) on a bf16 channels-last map (C % 8 == 0).
 * tspn_stem_pool_bf16: BasicStem.forward whole (conv + FrozenBN + ReLU, then max_pool2d(3, 2, 1)) in one conv
 * launch: out bf16 [NB,PH,PW,Cout], PH = (OH - 1) / 2 + 1; bit-identical to tspn_max_pool_nhwc_bf16(3, 2, 1) of
 * tspn_stem_conv_bf16, whose map it never writes.  Same operands and workspace as tspn_stem_conv_bf16. */
size_t tspn_stem_bf16_workspace_bytes(int64_t NB, int64_t H, int64_t W);
int tspn_pack_stem_bf16(const float* w, int64_t Cout, uint16_t* frag, void* stream);
int tspn_stem_conv_bf16(const float* x, int64_t NB, int64_t H, int64_t W, const uint16_t* frag, int64_t Cout,
                        const float* bias, void* workspace, size_t workspace_bytes, uint16_t* out, void* stream);
int tspn_max_pool_nhwc_bf16(const uint16_t* x, int64_t NB, int64_t H, int64_t W, int64_t C, int64_t k,
                            int64_t stride, int64_t pad, uint16_t* out, void* stream);
int tspn_stem_pool_bf16(const float* x, int64_t NB, int64_t H, int64_t W, const uint16_t* frag, int64_t Cout,
                        const float* bias, void* workspace, size_t workspace_bytes, uint16_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TSPN_MI355X_H */
