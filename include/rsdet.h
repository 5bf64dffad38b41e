/* rsdet.h -- C ABI of librsdet_hip.so: the MI355X (gfx950) oriented-detection
 * hot path that replaces JDet's jt.code() native-op seam.
 *
 * Boundary rules (every entry point):
 *   - extern "C", plain pointers + sizes, int status return (0 = RSDET_OK);
 *   - all pointers are DEVICE pointers unless the name says host;
 *   - never allocates or frees device memory, never synchronises: outputs and
 *     workspaces are caller-provided, work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the null stream);
 *   - re-entrant, no global state; graph-capturable.
 *   - tensors are dense row-major fp32 unless stated.
 *
 * Each declaration cites the reference interface it replaces; paths are
 * relative to /root/reference/python/jdet/.  The reference bakes scalar
 * parameters (thresholds, conv geometry, BOX_LENGTH) into JIT source strings
 * (ops/nms_rotated.py:498-503, ops/dcn_v1.py:314-338); here they are arguments.
 */
#ifndef RSDET_H_
#define RSDET_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSDET_OK 0
#define RSDET_EINVAL (-22)  /* bad argument (shape, stride, null pointer, workspace too small) */
#define RSDET_ELAUNCH (-5)  /* HIP reported a launch error */

/* ABI version of this header (bumped on any signature change). */
int rsdet_abi_version(void);

/* ---- a1/a2  rotated IoU -------------------------------------------------------
 * Replaces box_iou_rotated / box_iou_rotated_v1:
 *   ops/box_iou_rotated.py:502-509 (jt.code call), kernel :413-461, CPU :487-500;
 *   ops/box_iou_rotated_v1.py:507-524.
 * boxes (n, stride>=5) rows = (cx, cy, w, h, theta[rad], ...); ious (n1, n2).
 * version 0 / 1 selects the vertex convention (v1 = y-down angle, _v1.py:69-72).
 * The v1 wrapper's "too small" post-filter (_v1.py:515-522) is host-side logic.
 * Workspace: rsdet_box_iou_rotated_ws_size(n1, n2_total, n2) bytes, 16-byte aligned: prepared
 * boxes (fp64 sincos once per box; n2_total = n2, or n_groups*n2 for per-group column sets) and
 * the sharded global work queue of overlapping pairs (about min(n1*n2, 4 Mi) x 16 B). */
size_t rsdet_box_iou_rotated_ws_size(int n1, long long n2_total, int n2);
int rsdet_box_iou_rotated_f32(const float* boxes1, int n1, int stride1, const float* boxes2,
                              int n2, int stride2, int version, float* ious, void* ws,
                              size_t ws_bytes, void* stream);

/* Batched form used by the assigner (one launch for all images of a batch):
 * rows [row_offsets[g], row_offsets[g+1]) of boxes1/ious belong to group g and
 * are compared with boxes2 + g*group_stride2 (floats; 0 = one shared column set, else n2*stride2).
 * row_offsets: n_groups+1 device ints; max_rows_per_group: host-known bound.
 * Replaces the per-image loop models/boxes/anchor_target.py:60-72 ->
 * assigner.py:94 -> iou_calculator.py:157-162. */
int rsdet_box_iou_rotated_grouped_f32(const float* boxes1, int n1, int stride1,
                                      const int* row_offsets, int n_groups,
                                      int max_rows_per_group, const float* boxes2, int n2,
                                      int stride2, long long group_stride2, int version,
                                      float* ious, void* ws, size_t ws_bytes, void* stream);

/* ---- a1/a4/a5/a6/a7  rotated IoU in one launch, and the fused (sparse) anchor targets ---------------
 * (csrc/anchor_target.hip)
 *
 * Prepared column sets: fp64 sincos once per box -> 40-byte prepared boxes + the bounding box of
 * every 64 consecutive bounding circles.  Cacheable by the caller: S2ANet's FAM anchor grid never
 * changes (s2anet_head.py:224-228 caches the grid itself for the same reason).  boxes (n_total,
 * stride>=5) = n_total / n_per_group slabs of n_per_group boxes (one slab per image for per-image
 * anchors).  prepared: rsdet_iou_prepared_bytes() bytes, 16-byte aligned. */
size_t rsdet_iou_prepared_bytes(long long n_total, int n_per_group);
int rsdet_iou_prepare_f32(const float* boxes, long long n_total, int n_per_group, int stride,
                          void* prepared, size_t prepared_bytes, void* stream);

/* Dense IoU matrix, ONE launch: every 16 x 256 tile is detected, zero-filled and clipped by the
 * workgroup that owns it (each element stored exactly once).  Same values as
 * rsdet_box_iou_rotated_f32 (same clipper), which keeps the three-launch form.
 * Replaces ops/box_iou_rotated.py:502-509 and the per-image loop anchor_target.py:60-72.
 * row_offsets (n_groups+1 device ints) groups the rows of boxes1 / ious; NULL = one group.
 * tile_table: optional n_row_tiles x {group, first row, rows (<=16), first row of the group} device
 * ints, built by a caller that knows the gt counts on the host (no empty workgroups); NULL = a
 * n_groups x ceil(max_rows_per_group / 16) grid with early exits.
 * prepared2 = rsdet_iou_prepare_f32 of the columns; per_group != 0: one column slab per group.
 * prepared1 = rsdet_iou_prepare_f32(boxes1, n1, n1, ...) or NULL (then every workgroup runs the fp64
 * sincos of its 16 rows itself).
 * heavy_from_col: performance hint, never correctness -- columns >= it hold LARGE boxes (the anchors of
 * the top pyramid levels overlap ~10x more gts); their tiles are cut into four 4-row sub-tiles so that
 * no single workgroup becomes the tail of the launch.  n2 (or any value >= n2): no such columns. */
int rsdet_box_iou_rotated_tiled_f32(const float* boxes1, int n1, int stride1, const int* row_offsets,
                                    int n_groups, int max_rows_per_group, const int* tile_table,
                                    int n_row_tiles, const void* prepared1, const void* prepared2, int n2,
                                    int per_group, int heavy_from_col, int version, float* ious,
                                    void* stream);

/* The same matrix in TWO launches -- the fastest dense form: (1) detection only (no clipper in the kernel: 12 KB of
 * LDS, 8 workgroups per CU) leaves per tile a survivor bit mask and its surviving pairs in a sharded global queue;
 * (2) every workgroup first zero-fills its share of the matrix SKIPPING the survivor bits, then clips its balanced share
 * of the queue underneath the draining stores.  No element is written twice.  Arguments as rsdet_box_iou_rotated_tiled_f32
 * plus: state = rsdet_box_iou_rotated_split_state_bytes() bytes that are zero on entry (left zero on exit), ws =
 * rsdet_box_iou_rotated_split_ws_size(n1, n2, n_row_tiles) bytes of scratch (n_row_tiles: the tile table's length, or
 * n_groups * ceil(max_rows_per_group / 16) without a table). */
size_t rsdet_box_iou_rotated_split_state_bytes(void);
size_t rsdet_box_iou_rotated_split_ws_size(int n1, int n2, int n_row_tiles);
int rsdet_box_iou_rotated_split_f32(const float* boxes1, int n1, int stride1, const int* row_offsets, int n_groups,
                                    int max_rows_per_group, const int* tile_table, int n_row_tiles,
                                    const void* prepared1, const void* prepared2, int n2, int per_group,
                                    int heavy_from_col, int version, float* ious, void* state, size_t state_bytes,
                                    void* ws, size_t ws_bytes, void* stream);

/* Dense IoU matrix in ONE launch with a TWO-TIER clipper (csrc/iou_fast.hip): values within the north star's
 * tolerance of the reference (|IoU - reference| <= 1e-4 asked, < 3e-6 measured), not bit-identical to it.
 * Tier 1 (every overlapping pair): intersection area by Green's theorem, one lane per pair, registers only.
 * Tier 2 (the reference-order clipper of rsdet_box_iou_rotated_f32): pairs in which a corner of one box lies within
 * 0.01 px of an edge of the other -- where the REFERENCE itself leaves the true area (its hull scan reads dist[] with
 * pre-sort indices, box_iou_rotated.py:199-212) -- pairs with IoU < 3e-5 (exact zeros are kept) and NaN boxes.
 * Replaces ops/box_iou_rotated.py:502-509 / box_iou_rotated_v1.py:507-524 for callers that need the VALUES only;
 * callers that derive indices from thresholds or ties (MaxIoUAssigner) keep the bit-exact entries above or use
 * rsdet_anchor_target_rotated_f32.  Arguments as rsdet_box_iou_rotated_tiled_f32, except that row tiles hold
 * rsdet_box_iou_rotated_fast_rows_per_tile() (= 32) rows (heavy column tiles: four 8-row sub-tiles). */
int rsdet_box_iou_rotated_fast_rows_per_tile(void);
int rsdet_box_iou_rotated_fast_f32(const float* boxes1, int n1, int stride1, const int* row_offsets, int n_groups,
                                   int max_rows_per_group, const int* tile_table, int n_row_tiles,
                                   const void* prepared1, const void* prepared2, int n2, int per_group,
                                   int heavy_from_col, int version, float* ious, void* stream);

/* anchor_target for a whole batch WITHOUT the (K, A) matrix: rotated IoU of the surviving pairs only
 * (1.2 % at S2ANet shapes) -> MaxIoUAssigner (column max / first argmax, thresholds, low-quality
 * rule with gt_max_assign_all = True: the LAST gt whose IoU equals its row maximum; a gt that
 * overlaps no anchor has row maximum 0 and, with min_pos_iou <= 0, claims every anchor) ->
 * PseudoSampler -> DeltaXYWHA encode of the positives -> labels / weights / counts.  Two launches
 * (tiles: detect + clip; column maxima by device-scope 64-bit atomic max, row maxima, and the few
 * pairs that equal their row's maximum inside the tile as entries; finish: one workgroup per
 * (256 anchors, image) applies the low-quality rule to those entries and writes the targets).
 * Replaces anchor_target.py:18-180, assigner.py:65-170, sampler.py:114-130, box_ops.py:184-230,
 * box_iou_rotated.py:502-509 for the batch.  Results equal the dense path's
 * (rsdet_box_iou_rotated_grouped_f32 -> rsdet_assign_wrt_overlaps_f32 -> rsdet_bbox2delta_rotated_f32)
 * bit for bit on finite boxes.
 *   gt_boxes (n1, stride1), gt_labels (n1) or NULL (then positives get label 1),
 *   anchors (n2, stride2) shared, or (n_groups, n2, stride2) when per_group != 0; prepared2 as above;
 *   prepared_gt = rsdet_iou_prepare_f32(gt_boxes, n1, n1, ...) or NULL (prepared inside);
 *   valid (n_groups, n2) uint8 or NULL: anchors outside it are ignored (gt_inds -1, weights 0) and
 *   take no part in the row maxima (anchor_target.py:124-130 subsets them away);
 *   tile_table / n_row_tiles / heavy_from_col as for rsdet_box_iou_rotated_tiled_f32; group_tile0
 *   (n_groups+1 device ints: first row tile of each group) goes with tile_table: both or neither.
 * Outputs, each optional (NULL): gt_inds / max_overlaps / labels / label_weights (n_groups, n2),
 * bbox_targets / bbox_weights (n_groups, n2, 5), totals[2] = { sum_g max(#pos_g, 1),
 * sum_g max(#neg_g, 1) } (anchor_target.py:79-80).
 *   two_tier != 0 (round 3; opt-in, the Python op's default is 0): every overlapping pair gets the Green-integral IoU of
 *          rsdet_geom_fast.h (one lane per pair; |error| < 3e-6 against the reference, budget 2e-5) and the
 *          reference-order clipper runs only where a DECISION could depend on the difference: pairs in the
 *          reference's fragile zone / IoU < 1e-6 / NaN, per gt the pairs within 4e-5 of its best value in a tile
 *          (-> exact row maxima and exact `== row maximum` matches of the low-quality rule), per anchor the gts
 *          within 4e-5 of its best value when there are several or when a threshold lies within 2e-5 of it.
 *          gt_inds, labels, weights, targets and totals are those of two_tier == 0 bit for bit; max_overlaps is
 *          the fast value (within the budget) wherever no decision needed the exact one.
 *   state: rsdet_anchor_target_rotated_state_bytes(n1, n2, n_groups) bytes that MUST be zero on entry;
 *          the call leaves them zero again (counters reset by its last workgroups), so one zeroed
 *          buffer serves every later call on the same stream.
 *   ws:    rsdet_anchor_target_rotated_ws_size(n2, n_groups, n_row_tiles) bytes of scratch (worst
 *          case sized: one 32 KiB entry slice per 16 x 256 tile; only the used part is touched). */
size_t rsdet_anchor_target_rotated_state_bytes(int n1, int n2, int n_groups);
size_t rsdet_anchor_target_rotated_ws_size(int n2, int n_groups, int n_row_tiles);
int rsdet_anchor_target_rotated_f32(
    const float* gt_boxes, int n1, int stride1, const int* gt_labels, const int* row_offsets, int n_groups,
    int max_rows_per_group, const int* tile_table, int n_row_tiles, const int* group_tile0, const float* anchors,
    int n2, int stride2, int per_group, const void* prepared2, const void* prepared_gt, int heavy_from_col,
    const unsigned char* valid, int version, int two_tier, float pos_iou_thr, float neg_iou_lo, float neg_iou_hi,
    float min_pos_iou, int match_low_quality,
    int labels_filled, float pos_weight, int reg_decoded_bbox, const float* means_host, const float* stds_host,
    int* gt_inds, float* max_overlaps, int* labels, float* label_weights, float* bbox_targets, float* bbox_weights,
    float* totals, void* state, size_t state_bytes, void* ws, size_t ws_bytes, void* stream);

/* ---- a15  focal + smooth-L1 of one S2ANet module (FAM or ODM), all pyramid levels, one pass each way ----
 * (csrc/losses.hip)  Replaces FocalLoss / sigmoid_focal_loss (models/losses/focal_loss.py:5-96), SmoothL1Loss
 * (models/losses/smooth_l1_loss.py:5-54) and the per-level reshapes of loss_fam_single / loss_odm_single
 * (models/roi_heads/s2anet_head.py:430-508): 2 x n_levels loss calls of ~10 elementwise kernels each.
 * cls_maps / box_maps: HOST arrays of n_levels device pointers to the prediction maps as the convolutions
 * leave them, (B, C, H_l, W_l) and (B, 5, H_l, W_l), fp32 or (bf16_maps != 0) bf16; hw_host[l] = H_l * W_l.
 * labels (B, A) int32 1-based (0 = background), label_weights (B, A), bbox_targets / bbox_weights (B, A, 5)
 * with A = sum_l hw[l], anchors level-major (the layout rsdet_anchor_target_rotated_f32 writes).
 * avg_factor: one device float (num_total_pos).  losses (2, n_levels): row 0 = w_cls * focal sums / avg,
 * row 1 = w_box * smooth-L1 sums / avg -- deterministic (fixed summation order).  alpha < 0: no alpha weighting.
 * ws: rsdet_s2a_loss_ws_size() bytes whose first word is zero on entry (left zero again on exit).
 * backward: grad_losses (2, n_levels) upstream; grad_cls / grad_box: host arrays of device pointers with the
 * layouts and storage type of the inputs, every element written. */
size_t rsdet_s2a_loss_ws_size(const int* hw_host, int n_levels, int B);
int rsdet_s2a_loss_forward(const void* const* cls_maps, const void* const* box_maps, int bf16_maps,
                           const int* hw_host, int n_levels, int B, int C, const int* labels,
                           const float* label_weights, const float* bbox_targets, const float* bbox_weights,
                           const float* avg_factor, float alpha, float gamma, float beta, float w_cls, float w_box,
                           float* losses, void* ws, size_t ws_bytes, void* stream);
int rsdet_s2a_loss_backward(const void* const* cls_maps, const void* const* box_maps, int bf16_maps,
                            const int* hw_host, int n_levels, int B, int C, const int* labels,
                            const float* label_weights, const float* bbox_targets, const float* bbox_weights,
                            const float* avg_factor, const float* grad_losses, float alpha, float gamma, float beta,
                            float w_cls, float w_box, void* const* grad_cls, void* const* grad_box, void* stream);

/* ---- a16  rotated NMS -----------------------------------------------------------
 * Replaces nms_rotated_cpu / nms_rotated_cuda: ops/nms_rotated.py:495-512
 * (kernels :353-411 + host sweep :450-493, CPU loop :414-449).
 * dets (n, box_len) with box_len 5, or 6 (6th column = class label: boxes of
 * different labels never suppress each other, :283-286).  order (n) int32 =
 * indices by descending score -- or any order that keeps the boxes of every label
 * score-descending (e.g. label-major: tiles whose two 64-box blocks hold disjoint
 * label ranges are skipped; if the caller PROMISES a label-major order by OR-ing
 * RSDET_NMS_LABEL_MAJOR into `ge`, the runs of equal labels are also swept
 * concurrently, one workgroup each).  A box is suppressed when its IoU with an earlier
 * kept box is >= thr (CPU-path predicate :444; ge=0 selects the CUDA-path `>` :403).
 * keep (n) uint8, indexed like dets (jt.where(keep) gives ascending indices, :525).
 * Workspace: rsdet_nms_rotated_ws_size(n) bytes, 16-byte aligned, device. */
#define RSDET_NMS_GE 1          /* suppress on IoU >= thr (reference CPU path); 0: IoU > thr (CUDA path) */
#define RSDET_NMS_LABEL_MAJOR 2 /* `order` is label-major: equal labels contiguous, score-descending inside */
size_t rsdet_nms_rotated_ws_size(int n);
int rsdet_nms_rotated_f32(const float* dets, int n, int box_len, const int* order, float thr,
                          int ge, uint8_t* keep, void* ws, size_t ws_bytes, void* stream);

/* ---- a20  horizontal-box NMS of the Oriented-RCNN proposal stage -------------------------------
 * Replaces Jittor's built-in jt.nms (third-party, un-vendored; call sites
 * models/roi_heads/oriented_rpn_head.py:219, ops/nms.py:9,44).  boxes_sorted (n,4) = (x1,y1,x2,y2)
 * ALREADY in descending-score order; IoU uses the legacy "+1" convention when plus_one != 0;
 * a box is suppressed when IoU > thr with an earlier kept box.  keep_sorted (n) uint8, by sorted
 * position.  Workspace rsdet_nms_hbb_ws_size(n), 16-byte aligned. */
size_t rsdet_nms_hbb_ws_size(int n);
/* Horizontal-box IoU (iof = 0) / IoF (iof = 1) matrix, not aligned: models/boxes/iou_calculator.py:164-257 (bbox_overlaps,
 * the default calculator of MaxIoUAssigner -- the Oriented RPN's assignment of configs/orcnn) in one pass, bit-identical to
 * the tensor form.  boxes: rows of `stride` >= 4 floats (x1, y1, x2, y2, ...); out (n1, n2). */
int rsdet_bbox_overlaps_f32(const float* boxes1, int n1, int stride1, const float* boxes2, int n2, int stride2, int iof,
                            float eps, float* out, void* stream);
int rsdet_nms_hbb_sorted_f32(const float* boxes_sorted, int n, float thr, int plus_one,
                             uint8_t* keep_sorted, void* ws, size_t ws_bytes, void* stream);

/* ---- a4  MaxIoUAssigner.assign_wrt_overlaps ---------------------------------------
 * Replaces models/boxes/assigner.py:111-170 (incl. the per-gt Python loop :151-160)
 * for a whole batch.  overlaps (n1, A) as produced by the grouped IoU; group g
 * owns rows [row_offsets[g], row_offsets[g+1]).  Outputs per group g, each (A):
 *   gt_inds[g*A + j]  -1 ignore / 0 negative / (row - row_offsets[g]) + 1
 *   max_overlaps[g*A + j]
 *   labels[g*A + j]   gt_labels[row] for positives else labels_filled (NULL gt_labels: skipped)
 * Tie rule: first index of the maximum.  gt_max_assign_all semantics: every
 * column whose IoU EQUALS the row maximum is given to that row, rows applied in
 * ascending order (later rows overwrite).  A group with no rows gets gt_inds = 0
 * (all negative) and max_overlaps = 0; the reference raises there
 * (assigner.py:91-92) and the host-side mirror keeps that raise.
 * Workspace: rsdet_assign_ws_size(n1) bytes. */
size_t rsdet_assign_ws_size(int n1);
int rsdet_assign_wrt_overlaps_f32(const float* overlaps, int n1, int A, const int* row_offsets,
                                  int n_groups, int max_rows_per_group, float pos_iou_thr,
                                  float neg_iou_lo, float neg_iou_hi, float min_pos_iou,
                                  int match_low_quality, int gt_max_assign_all,
                                  const int* gt_labels, int labels_filled, int* gt_inds,
                                  float* max_overlaps, int* labels, void* ws, size_t ws_bytes,
                                  void* stream);

/* ---- a7/a9  DeltaXYWHA box coder -----------------------------------------------------
 * Replaces bbox2delta_rotated / delta2bbox_rotated: models/boxes/box_ops.py:184-230,
 * :233-289 (norm_angle 'le135' :176-182).  All arrays (n, 5).  means/stds: 5 HOST floats.
 * max_ratio = |log(wh_ratio_clip)| (16/1000 default, 1e-6 in bbox_decode
 * models/roi_heads/s2anet_head.py:631-654). */
int rsdet_bbox2delta_rotated_f32(const float* proposals, const float* gt, int n,
                                 const float* means_host, const float* stds_host, float* deltas,
                                 void* stream);
int rsdet_delta2bbox_rotated_f32(const float* rois, const float* deltas, int n,
                                 const float* means_host, const float* stds_host,
                                 float max_ratio, float* boxes, void* stream);

/* ---- a9+a10  fused bbox_decode + AlignConv.get_offset ---------------------------------
 * Replaces bbox_decode (s2anet_head.py:631-654) followed by AlignConv.get_offset
 * (:676-713) and their per-image Python loops (:646-653, :717-720).
 * bbox_pred (B,5,H,W) NCHW deltas; anchors (H*W,5); outputs
 * refined (B,H,W,5) and offset (B, 2*ks*ks, H, W) ordered (y,x) per tap. */
int rsdet_s2a_refine_and_offset_f32(const float* bbox_pred, const float* anchors, int B, int H,
                                    int W, float stride_px, int ks, const float* means_host,
                                    const float* stds_host, float max_ratio, float* refined,
                                    float* offset, void* stream);
/* The same for SEVERAL pyramid levels in one launch (the per-level loop of s2anet_head.py:229-239); pred may be bf16
 * (pred_bf16 != 0: the prediction of an autocast step, widened exactly -- no cast pass in front).  pred[l] (B,5,H,W)
 * contiguous; means / stds: 5 host floats or NULL (0 / 1). */
#define RSDET_S2A_MAX_LEVELS 8
typedef struct rsdet_s2a_levels {
  int n_levels, B, ks, pred_bf16;
  int H[RSDET_S2A_MAX_LEVELS], W[RSDET_S2A_MAX_LEVELS];
  float stride[RSDET_S2A_MAX_LEVELS];
  const void* pred[RSDET_S2A_MAX_LEVELS];
  const float* anchors[RSDET_S2A_MAX_LEVELS];
  float* refined[RSDET_S2A_MAX_LEVELS]; /* (B,H,W,5) or NULL */
  float* offset[RSDET_S2A_MAX_LEVELS];  /* (B,2*ks*ks,H,W) or NULL */
  const float* means;
  const float* stds;
  float max_ratio;
} rsdet_s2a_levels;
int rsdet_s2a_refine_and_offset_multi(const rsdet_s2a_levels* levels, void* stream);

/* ---- a12  Active Rotating Filter ---------------------------------------------------------
 * Replaces arf_forward / arf_backward: ops/orn.py:260-280 (kernels :17-72, CPU :138-211).
 * weight (O, I, nOri, kH, kW); indices (nOri*kH*kW, nRot) uint8, 1-based;
 * out / grad_out (O*nRot, I*nOri, kH, kW); grad_weight like weight. */
int rsdet_arf_forward_f32(const float* weight, const uint8_t* indices, int O, int I, int nOri,
                          int kH, int kW, int nRot, float* out, void* stream);
int rsdet_arf_backward_f32(const uint8_t* indices, const float* grad_out, int O, int I, int nOri,
                           int kH, int kW, int nRot, float* grad_weight, void* stream);

/* RotationInvariantPooling (ops/orn.py:595-617 of the reference): y[n,f,p] = max_k x[n, f*nOri + k, p] over the nOri
 * orientation channels; x (N, F*nOri, H, W), y (N, F, H, W), both NCHW (nhwc = 0) or both channels-last (nhwc = 1);
 * bf16 != 0: 2-byte bfloat16 elements, else float.  Backward: the gradient is shared equally by tied maxima (the rule of
 * the torch.amax this replaces). */
int rsdet_ori_maxpool_forward(const void* x, int bf16, int N, int F, int nOri, int HW, int nhwc, void* y, void* stream);
int rsdet_ori_maxpool_backward(const void* x, const void* grad_y, int bf16, int N, int F, int nOri, int HW, int nhwc,
                               void* grad_x, void* stream);

/* ---- 8(f)4  rotation-invariant encoding ------------------------------------------------------
 * Replaces rie_forward / rie_backward: ops/orn.py:516-540 (CPU kernels :290-363).  feature / aligned /
 * grad (nBatch, nFeature*nOri) fp32 [H = W = 1]; direction (nBatch, nFeature) uint8 = first arg-max over
 * the orientations; aligned[(l - direction + nOri) % nOri] = feature[l]; backward is the inverse shift. */
int rsdet_rie_forward_f32(const float* feature, int nBatch, int nFeature, int nOri, uint8_t* direction,
                          float* aligned, void* stream);
int rsdet_rie_backward_f32(const uint8_t* direction, const float* grad_out, int nBatch, int nFeature, int nOri,
                           float* grad_in, void* stream);

/* ---- a11  deformable convolution v1 (AlignConv) ---------------------------------------------
 * Replaces deformable_im2col / deformable_col2im / deformable_col2im_coord:
 * ops/dcn_v1.py:309-410 (kernels :132-306).  Geometry = the reference's argument list.
 * im (B,C,H,W); offset (B, dg*2*kh*kw, Ho, Wo); col (C*kh*kw, B, Ho, Wo).
 * col2im ACCUMULATES into grad_im with float atomics (zero it first, :405);
 * col2im_coord writes grad_offset (B, dg*2*kh*kw, Ho, Wo). */
typedef struct rsdet_dcn_geom {
  int C, H, W;    /* input channels / height / width */
  int kh, kw;     /* kernel */
  int ph, pw;     /* padding */
  int sh, sw;     /* stride */
  int dh, dw;     /* dilation */
  int B;          /* images in this call (the reference's parallel_imgs / im2col_step) */
  int dg;         /* deformable groups */
} rsdet_dcn_geom;
int rsdet_deform_im2col_f32(const float* im, const float* offset, const rsdet_dcn_geom* g,
                            float* col, void* stream);
int rsdet_deform_col2im_f32(const float* col, const float* offset, const rsdet_dcn_geom* g,
                            float* grad_im, void* stream);
int rsdet_deform_col2im_coord_f32(const float* col, const float* im, const float* offset,
                                  const rsdet_dcn_geom* g, float* grad_offset, void* stream);

/* Channels-last forms of the two kernels above (same arithmetic, MI355X-first layout):
 * im / grad_im (B,H,W,C); colT (B*Ho*Wo, kh*kw, C) so that the convolution is the plain GEMM
 * out(B*Ho*Wo, O) = colT x W(kh*kw*C, O) with NHWC output; offset stays (B, dg*2*kh*kw, Ho, Wo).
 * One wave per output position, lanes over channels: every access, incl. the fp32 atomics of
 * col2im, is a contiguous 256-B / 1-KiB segment per wave-instruction. */
int rsdet_deform_im2col_nhwc_f32(const float* im, const float* offset, const rsdet_dcn_geom* g,
                                 float* colT, void* stream);
int rsdet_deform_col2im_nhwc_f32(const float* colT, const float* offset, const rsdet_dcn_geom* g,
                                 float* grad_im, void* stream);
/* Gather form of the channels-last col2im for ONE deformable group: the (position, tap, corner) -> input pixel map
 * is inverted first (integer histogram / scan / fill in ws), then every input pixel sums its contributions from
 * colT with plain loads and ONE store per element -- no floating-point atomics, grad_im need not be zeroed. */
size_t rsdet_deform_col2im_gather_ws_size(const rsdet_dcn_geom* g);
int rsdet_deform_col2im_gather_nhwc_f32(const float* colT, const float* offset, const rsdet_dcn_geom* g,
                                        float* grad_im, void* ws, size_t ws_bytes, void* stream);

/* The same index for SEVERAL calls in one go -- the five AlignConv levels of a step (models/roi_heads/s2anet_head.py:
 * 603-660 calls DeformConv once per level; each backward, dcn_v1.py:456-556, scatters on its own): pixels and items of
 * the levels are laid end to end, ONE histogram / scan / fill (5 launches in all) inverts every level's map, then each
 * level gathers with rsdet_deform_col2im_gather_indexed_nhwc_*.  index_multi writes, for the host, pix_base[0..n]
 * (level l's start array = (int*)((char*)ws + aligned start offset) -- returned through the level's own pointer below)
 * and the byte offsets of the shared ent_row / ent_w arrays inside ws.  A level's `start` argument of the gather is
 * (const int*)((char*)ws + start_offset) + pix_base[l] with start_offset = ((n_pix_total + 1) * 4 rounded up to 256).
 * Same results as the per-call form up to the order of the fp32 additions inside one pixel (entries of a pixel are
 * filed in arrival order in both forms). */
#define RSDET_DCN_INDEX_MAX_LEVELS 8
typedef struct rsdet_dcn_index_levels {
  int n_levels;
  const float* offset[RSDET_DCN_INDEX_MAX_LEVELS]; /* (B, 2*kh*kw, Ho, Wo) of each level */
  rsdet_dcn_geom geom[RSDET_DCN_INDEX_MAX_LEVELS]; /* dg must be 1 */
} rsdet_dcn_index_levels;
size_t rsdet_deform_col2im_index_multi_ws_size(const rsdet_dcn_index_levels* levels);
int rsdet_deform_col2im_index_multi_f32(const rsdet_dcn_index_levels* levels, void* ws, size_t ws_bytes,
                                        long long* pix_base, size_t* ent_row_offset, size_t* ent_w_offset,
                                        void* stream);
int rsdet_deform_col2im_gather_indexed_nhwc_f32(const float* colT, const rsdet_dcn_geom* g, const int* start,
                                                const int* ent_row, const float* ent_w, float* grad_im, void* stream);
int rsdet_deform_col2im_gather_indexed_nhwc_bf16col_f32(const uint16_t* colT, const rsdet_dcn_geom* g,
                                                        const int* start, const int* ent_row, const float* ent_w,
                                                        float* grad_im, void* stream);
/* ... grad_im (B,H,W,C) written as bf16, round to nearest even: the gradient of a bf16 input (autocast step). */
int rsdet_deform_col2im_gather_indexed_nhwc_bf16col_bf16(const uint16_t* colT, const rsdet_dcn_geom* g,
                                                         const int* start, const int* ent_row, const float* ent_w,
                                                         uint16_t* grad_im, void* stream);

/* bf16 column matrices for the autocast step (BASELINE configs 2 / 4): the products that consume / produce them run
 * on bf16 MFMA through rocBLAS, the images, offsets, interpolation weights, sums and grad_im stay fp32.
 * im2col_bf16col: same as rsdet_deform_im2col_f32 (dcn_v1.py:309-339) with col stored as bf16 (round to nearest
 * even); only the AlignConv geometry (3x3 taps, channels per deformable group a multiple of 16), else RSDET_EINVAL.
 * col2im_gather_nhwc_bf16col: same as rsdet_deform_col2im_gather_nhwc_f32 with colT read as bf16. */
int rsdet_deform_im2col_bf16col_f32(const float* im, const float* offset, const rsdet_dcn_geom* g, uint16_t* col,
                                    void* stream);
int rsdet_deform_col2im_gather_nhwc_bf16col_f32(const uint16_t* colT, const float* offset, const rsdet_dcn_geom* g,
                                                float* grad_im, void* ws, size_t ws_bytes, void* stream);

/* AlignConv as an implicit GEMM on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16): replaces
 * DeformConvFunction.forward, dcn_v1.py:412-454 (im2col :309-339 + the product :448-452), for the geometry AlignConv
 * uses (3x3, stride 1, dilation 1, one deformable group; models/roi_heads/s2anet_head.py:603-660) in the arithmetic of
 * a bf16 autocast step; the column matrix is not written unless asked for.
 *   im_nhwc (B,H,W,C) bf16; offset (B,18,Ho,Wo) fp32; weight (O, 9*C) bf16 with k = tap*C + c;
 *   out bf16: out_nhwc ? (B,Ho,Wo,O) : (B,O,Ho,Wo); colT: NULL, or (B*Ho*Wo, 9*C) bf16 = the sampled columns
 *   (saved for the weight gradient).  _supported: 1 when the geometry is covered (C % 64 == 0, C <= 2048, O % 32 == 0,
 *   B*H*W*C < 2^31), else 0 and the entry point returns RSDET_EINVAL.
 * _f32: the same kernel in exact fp32 (v_mfma_f32_32x32x2_f32; C % 32 == 0, C <= 1024): im_nhwc / weight / out / colT
 * fp32, the sampled columns bit-identical to rsdet_deform_im2col_f32 (same operation order, no contraction). */
int rsdet_alignconv_mfma_supported(const rsdet_dcn_geom* g, int O);
int rsdet_alignconv_mfma_f32_supported(const rsdet_dcn_geom* g, int O);
int rsdet_alignconv_fwd_mfma_f32(const float* im_nhwc, const float* offset, const float* weight,
                                 const rsdet_dcn_geom* g, int O, int out_nhwc, float* out, float* colT, void* stream);

/* 3x3 / stride 1 / padding 1 convolution of a channels-last bf16 map as an implicit GEMM on the matrix cores
 * (csrc/conv3x3_mfma.hip): one workgroup per image row x 256 output channels, both operands by LDS-DMA, bf16 products,
 * fp32 accumulation.  The library kernel behind the shared-weight tower convolutions of the S2ANet head
 * (models/roi_heads/s2anet_head.py:127-186 of the reference builds them as ConvModule(256, 256, 3)); their
 * backward-data is the same call on the flipped, transposed weights.
 * x (B, H, W, C); weight (O, 3, 3, C) = the storage of a channels_last (O, C, 3, 3) tensor; out (B, H, W, O).
 * Optional fused epilogue: bias (O fp32, NULL = none), relu, live (H*W bytes shared by all images, NULL = all live): positions whose byte
 * is 0 -- the gap pixels of the pyramid canvas -- are written as zeros.  C % 64 == 0, O % 32 == 0. */
int rsdet_conv3x3_mfma_supported(int B, int H, int W, int C, int O);
/* 1x1 convolution + frozen-statistics BatchNorm + residual add + ReLU of a channels-last bf16 map as ONE launch
 * (csrc/gemm1x1_mfma.hip: the conv3x3 kernel's tile and fragment machinery at K = C, the per-channel affine map, the
 * residual and the activation in the epilogue; the convolution's raw output never reaches memory).  Replaces conv1 /
 * conv3 / downsample + bn (+ identity) + relu of the reference's Bottleneck, models/backbones/resnet.py:57-93, in the
 * norm_eval mode of :177-184.  x (M, K), weight (N, K), residual / out (M, N) row-major bf16, M = B*H*W positions;
 * running_mean / running_var / gamma / beta fp32 (N) -- mean == var == NULL: a plain convolution with `beta` as its
 * bias; gamma NULL: 1; beta NULL: 0.  K % 64 == 0, N % 32 == 0.
 * Its backward through the BatchNorm: rsdet_bn_gate_sums_nhwc_bf16 (gate + grad_beta sums); the BatchNorm's scale rides
 * in the weights of the backward-data GEMM and in the fold of the weight gradient, and grad_gamma comes from that fold's
 * row dots (rsdet_bn_affine_grads_finish_multi_f32) -- exact for any gamma, zero included. */
int rsdet_gemm1x1_mfma_supported(long long M, int N, int K);
int rsdet_conv1x1_bn_act_fwd_bf16(const uint16_t* x, const uint16_t* weight, long long M, int N, int K,
                                  const float* running_mean, const float* running_var, const float* gamma,
                                  const float* beta, float eps, const uint16_t* residual, int relu, uint16_t* out,
                                  void* stream);
/* out[i] = sum_s partial[s][i], fp32 sums in slab order, stored fp32 or rounded once to bf16 (out_bf16): the fold of the
 * split-K 1x1 weight gradient (ops/conv1x1.py; the reference's cuDNN weight gradient, resnet.py:101-126 backward).
 * n % 4 == 0. */
int rsdet_sum_slabs_f32(const float* partial, int S, long long n, void* out, int out_bf16, void* stream);
/* ... with row r of the (n / row_len, row_len) result multiplied by gamma[r] / sqrt(running_var[r] + eps) before the
 * rounding: the weight gradient of a convolution that feeds an eval-mode BatchNorm, formed from the gradient of the
 * BatchNorm's OUTPUT (ops/bottleneck.py, ops/conv_bn.py).  row_len % 4 == 0, n % row_len == 0; gamma NULL: 1.
 * weight (bf16, the result's shape: the convolution's own weight) and rowdot (n / row_len floats), both or neither:
 * rowdot[r] = sum_i weight[r][i] U[r][i], U the unscaled fp32 sum = sum_p gz[p, r] conv[p, r] -- the term the
 * BatchNorm's scale gradient needs (the reference's batch-norm backward, resnet.py:57-93 under norm_eval). */
int rsdet_sum_slabs_rowscale_f32(const float* partial, int S, long long n, int row_len, const float* running_var,
                                 const float* gamma, float eps, const uint16_t* weight, float* rowdot, void* out,
                                 int out_bf16, void* stream);
/* out (C, O) = transpose of weight (O, C), column o scaled by gamma[o] / sqrt(running_var[o] + eps) (running_var NULL:
 * plain transpose): the weight operand of rsdet_conv1x1_dgrad_bf16. */
int rsdet_weight_transpose_scale_bf16(const uint16_t* weight, int O, int C, const float* running_var, const float* gamma,
                                      float eps, uint16_t* out, void* stream);
/* Backward-data of a 1x1 convolution through the streaming GEMM of rsdet_conv1x1_bn_act_fwd_bf16, with the NEXT backward
 * step of the Bottleneck (/root/reference/python/jdet/models/backbones/resnet.py:57-93, backward of :80-91) in its
 * epilogue: grad_in[p, c] = epi(sum_o grad_out[p, o] wt[c, o]); grad_out (M, O), wt (C, O), grad_in / side (M, C) bf16.
 *   mode 0: identity.
 *   mode 2: the convolution's input was side = relu(bn(.)) of an eval-mode BatchNorm over its C channels:
 *           grad_in = [side > 0] acc -- the gated gradient of THAT BatchNorm's OUTPUT (its scale rides in the weights of
 *           the next backward step) -- and, with ws (rsdet_conv1x1_dgrad_ws_size bytes), its per-channel sums: folded
 *           into grad_beta[c] = sum_p grad_in[p, c] when grad_beta is given, else left in ws as (C, S, 2) floats ([0]
 *           the sum, [1] zero; S = rsdet_conv1x1_dgrad_slices) for rsdet_bn_affine_grads_finish_multi_f32.
 *   mode 3: grad_in = acc + side (the gradient arriving through the identity branch).
 * C % 32 == 0, O % 64 == 0 (rsdet_gemm1x1_mfma_supported(M, C, O)). */
size_t rsdet_conv1x1_dgrad_ws_size(long long M, int C, int O);
int rsdet_conv1x1_dgrad_slices(long long M, int C, int O);
int rsdet_conv1x1_dgrad_bf16(const uint16_t* grad_out, const uint16_t* wt, long long M, int C, int O, int mode,
                             const uint16_t* side, float* grad_beta, void* ws, size_t ws_bytes, uint16_t* grad_in,
                             void* stream);
/* Gate pass of the fused convolution + eval BatchNorm (+ identity) + ReLU nodes, whose pre-BatchNorm value is never
 * stored: grad_z = grad_y [y > 0] (relu != 0; relu == 0: grad_z must be NULL, nothing is written) and the per-slice
 * channel sums of grad_z left in ws as (C, S, 2) floats ([0] the sum, [1] zero; S = rsdet_bn_gate_sums_nhwc_slices; ws of
 * rsdet_bn_act_backward_nhwc_ws_size bytes, NULL = no sums).  bf16 channels-last, C / 8 a divisor of 256. */
int rsdet_bn_gate_sums_nhwc_slices(int N, int C, int HW);
int rsdet_bn_gate_sums_nhwc_bf16(const uint16_t* grad_y, const uint16_t* y, int N, int C, int HW, int relu,
                                 uint16_t* grad_z, void* ws, size_t ws_bytes, void* stream);
/* The affine-parameter gradients of up to 4 eval-mode BatchNorms behind convolutions as ONE launch (the three of a
 * Bottleneck's backward, ops/bottleneck.py).  Job j: partial[j] (C, S, 2) sums from the two passes above; rowdot[j] from
 * the weight-gradient folds (rsdet_sum_slabs_rowscale_f32, rsdet_conv3x3_wrw_mfma_rowscale_bf16):
 *   grad_beta[j][c] = sum_s partial[c][s][0];  grad_gamma[j][c] = (rowdot[c] - mean[c] grad_beta[c]) / sqrt(var[c] + eps)
 * = sum_p gz[p, c] xhat[p, c] with the convolution output the forward computed, for ANY gamma (the reference's
 * batch-norm backward under norm_eval, resnet.py:177-184).  rowdot[j] or grad_gamma[j] NULL: only grad_beta. */
int rsdet_bn_affine_grads_finish_multi_f32(int n, const float* const* partial, const int* C, const int* S,
                                           const float* const* rowdot, const float* const* running_mean,
                                           const float* const* running_var, const float* eps, float* const* grad_gamma,
                                           float* const* grad_beta, void* stream);
/* The 1x1 convolutions of a VAN block on NCHW fp32 maps as streaming GEMMs on the fp32 matrix cores, the block's
 * elementwise tails in their epilogues (csrc/van_gemm.hip; /root/reference/python/jdet/models/backbones/van.py:140-263:
 * Mlp.fc1 / fc2, AttentionModule.conv1, SpatialAttention.proj_1 / proj_2 and the expressions around them, Block.execute;
 * the backward-data GEMMs are the same call on transposed weights).  out (n_img, M, P) = epi(weight (M, K) . x (n_img, K, P)),
 * fp32, pixels contiguous; r = output row (channel), v* (M) per-row vectors, s* maps of the output's shape:
 *   epi 0  out0 = acc                              4  out0 = s0 v0[r] + acc v1[r] + v2[r] (+ s1 v3[r] if s1; v0 NULL: 1)
 *       1  out0 = acc + v0[r]                      5  out0 = acc s0, out1 = acc s1
 *       2  out0 = acc + v0[r], out1 = GELU(out0)   6  out0 = acc s0
 *       3  out0 = acc + v0[r], out1 = out0 s0
 * (GELU = the erf form of jt.nn.GELU.)  _supported: M a multiple of 160 / 128 (P % 64 == 0) or 64 (P % 128 == 0), K % 32 == 0;
 * every pointer 16-byte aligned.  Exact fp32: a k-ordered fmaf chain per output element. */
int rsdet_van_gemm_f32_supported(int M, int K, int P, int n_img);
int rsdet_van_gemm_f32(const float* weight, const float* x, int M, int K, int P, int n_img, int epi, const float* v0,
                       const float* v1, const float* v2, const float* v3, const float* s0, const float* s1, float* out0,
                       float* out1, void* stream);
/* Weight gradient of those convolutions: partial[s] (M, N) = sum over split s's (image, 32-pixel chunk) range of
 * g[img][m][p] x[img][n][p] (g (n_img, M, P), x (n_img, N, P)); S = rsdet_van_wgrad_f32_splits partials, summed in order by a
 * fold below.  M a multiple of 160 / 128 / 64, N of 64, P of 32. */
int rsdet_van_wgrad_f32_supported(int M, int N, int P, int n_img);
int rsdet_van_wgrad_f32_splits(int M, int N, int P, int n_img);
int rsdet_van_wgrad_f32(const float* g, const float* x, int M, int N, int P, int n_img, float* partial, void* stream);
/* Fold of the partials U = sum_s partial[s] (M, N), one launch per convolution, for a convolution whose output (plus
 * bias) is multiplied by a per-channel layer scale before it meets the gradient g (Block.execute, van.py:258-261):
 *   grad_w = row_scale[m] U[m][n] (row_scale NULL: 1);  grad_b[m] = row_scale[m] gs[m], gs = sum_p g[m, p] from the slice
 *   partials gs_tab[(m * gs_ns + j) * gs_stride];  grad_rs[m] (the layer scale's gradient) = sum_n w[m][n] U[m][n] +
 *   bias[m] gs[m] + sc[m] R2[m] + sh[m] R1[m]  (r_tab (M, r_ns, 2): slice partials of (sum_p g, sum_p g x) when the scaled sum
 *   also holds the shortcut xn = x sc + sh; NULL: absent).  grad_b / grad_rs NULL: not formed.  N % 4 == 0. */
typedef struct rsdet_van_rows_fold {
  const float *partial, *row_scale, *w, *gs_tab, *bias, *r_tab, *sc, *sh;
  float *grad_w, *grad_b, *grad_rs;
  int S, M, N, gs_ns, gs_stride, r_ns;
} rsdet_van_rows_fold;
int rsdet_van_fold_rows_f32(const rsdet_van_rows_fold* f, void* stream);
/* Fold for a convolution whose INPUT is a training-mode BatchNorm folded into its weights (Block.norm1 -> proj_1, norm2 ->
 * fc1): partial (S, K, O) are TRANSPOSED unscaled weight gradients against the raw input (rsdet_van_wgrad_f32(x, g)).
 * One launch produces grad_w (O, K) = sc[k] UT[k][o] + sh[k] gs[o], grad_b = gs, the BatchNorm's grad_gamma / grad_beta, and
 * the per-channel constants v0..v3 with which the backward-data GEMM's epilogue (rsdet_van_gemm_f32 epi 4, s1 = the
 * BatchNorm's input) applies the whole BatchNorm backward:  grad_x = s0 v0 + acc v1 + v2 + x v3.
 * gs_tab: slice partials of sum_p g[o, p] at [(o * gs_ns + j) * gs_stride]; r_tab / ls: the (sum_p G, sum_p G x) slice
 * partials (rsdet_van_chan_reduce_f32 mode 0) and layer scale of a second path into the BatchNorm's output (the
 * attention's own shortcut), or NULL; cnt = images * pixels. */
typedef struct rsdet_van_bn_fold {
  const float *partial, *wt, *gs_tab, *r_tab, *ls, *mean, *rstd, *sc, *sh;
  float *grad_w, *grad_b, *grad_gamma, *grad_beta, *v0, *v1, *v2, *v3;
  int S, K, O, gs_ns, gs_stride, r_ns;
  float cnt;
} rsdet_van_bn_fold;
int rsdet_van_fold_bn_f32(const rsdet_van_bn_fold* f, void* stream);
/* Per-channel reductions of NCHW fp32 maps (N, C, P), one workgroup per (plane, slice), deterministic:
 *   mode 0  tab[(c * ns + j) * 2 + {0, 1}] = slice partials of (sum_p a, sum_p a b)  (b NULL: 0)
 *   mode 1  the slice's (mean, sum of squared deviations from that mean) of a: BatchNorm statistics
 * ns = N * rsdet_van_chan_slices(P).  P % 4 == 0. */
int rsdet_van_chan_slices(int P);
int rsdet_van_chan_reduce_f32(const float* a, const float* b, int N, int C, int P, int mode, float* tab, void* stream);
/* Training-mode BatchNorm folded into the 1x1 convolution behind it (nn.BatchNorm2d semantics: batch statistics, biased
 * variance for the normalisation, running statistics updated with `momentum` and the unbiased variance;
 * van.py:258-261 norm1 / norm2): from the mode-1 slice statistics tab (K, ns, 2) of slices of len pixels,
 *   w_out[o][k] = w[o][k] sc[k],  b_out[o] = b[o] + sum_k w[o][k] sh[k],  sc = gamma rstd,  sh = beta - mean sc,
 * and mean / rstd / sc / sh (K each) for the backward.  running_* / num_batches_tracked (int64) NULL: not updated.
 * ls / b2 (K each; layer scale and bias of the LAST convolution of the half this BatchNorm opens) -> the constants of that
 * convolution's residual epilogue: e0 = 1 + ls sc, e2 = ls (b2 + sh) with the attention's shortcut (shortcut != 0), else
 * e0 = 1, e2 = ls b2. */
typedef struct rsdet_van_bn_prep {
  const float *tab, *gamma, *beta, *w, *b;
  float *w_out, *b_out, *mean, *rstd, *sc, *sh, *running_mean, *running_var;
  void* num_batches_tracked;
  const float *ls, *b2;
  float *e0, *e2;
  int shortcut;
  int O, K, ns, len;
  float eps, momentum;
} rsdet_van_bn_prep;
int rsdet_van_bn_prep_f32(const rsdet_van_bn_prep* f, void* stream);
/* n <= 5 weight transposes as one launch: dst[j] (K[j], O[j]) = transpose(src[j] (O[j], K[j])), row o of src scaled by
 * row_scale[j][o] (NULL: 1): the operands of a block's backward-data GEMMs. */
int rsdet_van_transposes_f32(int n, const float* const* src, const float* const* row_scale, float* const* dst, const int* O,
                             const int* K, void* stream);
/* A whole VAN Block (Block.execute, /root/reference/python/jdet/models/backbones/van.py:216-261, with Mlp :140-175,
 * AttentionModule :177-192 and SpatialAttention :195-213 inside it) forward and backward as ONE call each
 * (csrc/van_block.hip): the 13 + 25 launches of the pieces above into caller-provided fp32 arenas.  x / out / grad_out /
 * grad_x: (N, C, H, W) NCHW; hidden width R; both BatchNorms in training mode (batch statistics; rm / rv / nbt = running
 * mean / variance / num_batches_tracked (int64), updated as nn.BatchNorm2d does, NULL: not kept).
 *   saved    rsdet_van_block_saved_floats            what the backward reads again (written by the forward)
 *   scratch  rsdet_van_block_{forward,backward}_scratch_floats
 *   grads    rsdet_van_block_grad_floats: the parameter gradients, 16-byte aligned slices in the order
 *            g1 be1 wp1 bp1 wd5 bd5 wd7 bd7 wc1 bc1 wp2 bp2 ls1 g2 be2 wf1 bf1 wd3 bd3 wf2 bf2 ls2 (each rounded up to 4 floats)
 * grad_x NULL: not formed.  _supported: every GEMM / weight-gradient shape of the block is one the kernels tile. */
typedef struct rsdet_van_block {
  int N, C, H, W, R;
  const float *g1, *be1, *wp1, *bp1, *wd5, *bd5, *wd7, *bd7, *wc1, *bc1, *wp2, *bp2, *ls1, *g2, *be2, *wf1, *bf1, *wd3, *bd3,
      *wf2, *bf2, *ls2;
  float *rm1, *rv1;
  void* nbt1;
  float *rm2, *rv2;
  void* nbt2;
  float eps1, mom1, eps2, mom2;
} rsdet_van_block;
/* Up to three rsdet_van_fold_rows_f32 folds in ONE launch (the block node's row folds feed nothing in its backward's data
 * chain and are issued together behind it). */
int rsdet_van_fold_rows_multi_f32(const rsdet_van_rows_fold* jobs, int n, void* stream);
/* The depthwise weight gradient (rsdet_dwconv2d_backward_weight_f32) in two calls: _partial_f32 leaves the per-tile partial
 * rows in ws (rsdet_dwconv2d_backward_weight_ws_size bytes; N >= 1), rsdet_dwconv2d_wgrad_finish_multi_f32 sums the
 * partials of up to four layers (arrays of n entries: workspace, N, C, H, W, K, grad_weight, grad_bias or NULL) in one
 * launch -- fixed order, the same values as the one-call form. */
int rsdet_dwconv2d_backward_weight_partial_f32(const float* grad_y, const float* x, const float* in_bias, int N, int C, int H,
                                               int W, int K, int dilation, void* ws, size_t ws_bytes, void* stream);
int rsdet_dwconv2d_wgrad_finish_multi_f32(int n, const void* const* ws, const int* N, const int* C, const int* H,
                                          const int* W, const int* K, float* const* grad_weight, float* const* grad_bias,
                                          void* stream);
/* LayerNorm over the CHANNELS of an NCHW map.  Replaces the norm at the end of every VAN stage,
 * models/backbones/van.py:303-306 (`x.flatten(2).transpose(1, 2)` -> nn.LayerNorm(C) -> reshape / permute back): biased
 * variance, rsqrt(var + eps), affine per channel.  x / y / grad (N, C, HW) fp32; mean / rstd (N, HW) are kept for the
 * backward; backward: grad_x, and grad_gamma / grad_beta (either NULL: skipped) through ws
 * (rsdet_chan_layernorm_ws_size bytes).  C <= 512. */
int rsdet_chan_layernorm_supported(int N, int C, int HW);
size_t rsdet_chan_layernorm_ws_size(int N, int C, int HW);
int rsdet_chan_layernorm_forward_f32(const float* x, const float* gamma, const float* beta, int N, int C, int HW, float eps,
                                     float* y, float* mean, float* rstd, void* stream);
int rsdet_chan_layernorm_backward_f32(const float* grad_y, const float* x, const float* mean, const float* rstd,
                                      const float* gamma, int N, int C, int HW, float* grad_x, float* grad_gamma,
                                      float* grad_beta, void* ws, size_t ws_bytes, void* stream);
int rsdet_van_block_supported(const rsdet_van_block* b);
/* 1: the backward's weight gradients, their folds and the depthwise weight gradients run on a side stream beside the chain
 * that produces grad_x (joined before the call returns control of the buffers to `stream`); 0 (default: measured 2.5 %
 * slower on the Oriented R-CNN step, the seven cross-stream edges per block cost more than the overlap returns): everything
 * on `stream`.  Returns the previous setting; any other argument only queries. */
int rsdet_van_block_side_stream(int on);
size_t rsdet_van_block_saved_floats(const rsdet_van_block* b);
size_t rsdet_van_block_grad_floats(const rsdet_van_block* b);
size_t rsdet_van_block_forward_scratch_floats(const rsdet_van_block* b);
size_t rsdet_van_block_backward_scratch_floats(const rsdet_van_block* b);
int rsdet_van_block_forward_f32(const rsdet_van_block* b, const float* x, float* out, float* saved, float* scratch,
                                void* stream);
int rsdet_van_block_backward_f32(const rsdet_van_block* b, const float* x, const float* grad_out, const float* saved,
                                 float* scratch, float* grad_x, float* grads, void* stream);

/* ---- a20  the control path of the Oriented R-CNN heads (csrc/orpn.hip) ------------------------
 * RandomSampler on masks.  Replaces models/boxes/sampler.py:57-111 (BaseSampler.sample) + :132-180 (RandomSampler: a uniform
 * subset of min(#pos, num_pos) positives, then of the negatives up to `num`, neg_pos_ub honoured) in the fixed-size form of
 * the train step: "the k candidates with the largest draws" of one uniform draw per candidate (pri, float or double) -- a
 * uniform k-subset -- by an exact radix select, ties to the lower index.  Candidate i < k_gt is ground truth i (add_gt_as_
 * proposals: its gt index is i + 1); candidate k_gt + j is row j of gt_inds (n_props, int32: > 0 positive, 0 negative, < 0
 * neither), masked by valid (n_props uint8, NULL: all).  Outputs, `num` rows each (num <= 1024): inds (int64, into the
 * gt-extended list: positives in ascending order, then negatives in ascending order, unused slots 0), is_pos / val (uint8),
 * assigned (int64: the gt row of a positive, 0 elsewhere), counts (2 x int64: #pos, #neg).  No host synchronisation.
 * ws: rsdet_sample_masked_ws_size(num) bytes, 16-byte aligned. */
size_t rsdet_sample_masked_ws_size(int num);
int rsdet_sample_masked(const int32_t* gt_inds, const uint8_t* valid, int n_props, int k_gt, const void* pri, int pri_f64,
                        int num, int num_pos, float neg_pos_ub, int64_t* inds, uint8_t* is_pos, uint8_t* val,
                        int64_t* assigned, int64_t* counts, void* ws, size_t ws_bytes, void* stream);
/* MidpointOffsetCoder.decode (models/boxes/coder.py:372-433) with rectpoly2obb + regular_obb (ops/bbox_transforms.py:
 * 507-548): anchors (n, 4) hbb + deltas (n, 6) -> out (n, 5) obb, w >= h, theta in [-pi/2, pi/2).  means / stds: 6 host
 * floats (NULL: 0 / 1); max_ratio = |log(wh_ratio_clip)|.  obb2hbb: ops/bbox_transforms.py:572-578, (n, 5) -> (n, 4). */
int rsdet_midpoint_offset_decode_f32(const float* anchors, const float* deltas, int n, const float* means, const float* stds,
                                     float max_ratio, float* out, void* stream);
int rsdet_obb2hbb_f32(const float* obb, int n, float* out, void* stream);
/* The proposals of a batch.  Replaces models/roi_heads/oriented_rpn_head.py:135-222 (_get_bboxes_single per image: per level
 * the nms_pre best scores in stable order, decode, boxes not larger than min_size dropped, one horizontal NMS over all
 * levels with a per-level coordinate offset, the first nms_post survivors) for all images at once, in the fixed-size form
 * of the train step: out (n_img, nms_post, 6) = (x, y, w, h, theta, score) rows in the reference's order with zero rows
 * behind the last proposal, flags (n_img, nms_post) uint8 = which rows are proposals.  score[l]: the SIGMOID of the
 * classification map of level l laid out PIXEL-MAJOR, (n_img, H_l, W_l, A) -- the reference's flattening, whose index breaks
 * ties; reg[l]: (n_img, 6 A, H_l, W_l); anchors[l]: (H_l W_l A, 4) in the
 * reference's flattening (pixel-major); hw[l] = H_l W_l.  Supported: <= 7 levels, A hw[l] < 2^21, nms_pre <= 2048,
 * sum_l min(nms_pre, A hw[l]) <= 16384.  ws: rsdet_orpn_proposals_ws_size bytes, 16-byte aligned. */
typedef struct rsdet_orpn_levels {
  int n_img, n_levels, A, nms_pre, nms_post;
  float nms_thr, min_size, max_ratio;
  float means[6], stds[6];
  int hw[8];
  const float* score[8];
  const float* reg[8];
  const float* anchors[8];
} rsdet_orpn_levels;
int rsdet_orpn_proposals_supported(const rsdet_orpn_levels* d);
int rsdet_orpn_proposals_n(const rsdet_orpn_levels* d); /* rows that enter the NMS per image */
size_t rsdet_orpn_proposals_ws_size(const rsdet_orpn_levels* d);
int rsdet_orpn_proposals_f32(const rsdet_orpn_levels* d, float* out, uint8_t* flags, void* ws, size_t ws_bytes, void* stream);
/* The Oriented RPN's two losses evaluated on the SAMPLED anchors.  Replaces models/roi_heads/oriented_rpn_head.py:274-480
 * (_get_targets_single's four dense target maps per image, images_to_levels, loss_single per level) for the configuration the
 * reference trains: sigmoid classification with CrossEntropyLossForRcnn (losses/cross_entropy_loss.py:24-31), SmoothL1Loss
 * (losses/smooth_l1_loss.py:5-24) on MidpointOffsetCoder.encode targets (boxes/coder.py:334-370), avg_factor = sum over
 * the images of max(#pos, 1) + max(#neg, 1).  The dense maps are zero-weighted outside the samples, so the sums are the same.
 * cls[l] (n_img, A, H_l, W_l) logits, reg[l] (n_img, 6 A, H_l, W_l); anchors (total, 4) level-major, index pixel * A + a
 * inside a level; inside (int64) maps a sampled index to its row of `anchors` (NULL: identity); gt[b] (k_gt[b], 5) the
 * ground truth of image b as the head sees it (theta negated); inds / is_pos / val / assigned (n_img, num) and counts
 * (n_img, 2) from rsdet_sample_masked.  forward: losses (2, n_levels) = loss_cls per level then loss_bbox per level, each
 * already times its weight; rec: rsdet_orpn_loss_rec_floats(n_img, num) floats kept for the backward.  backward: cls[l] /
 * reg[l] are ZERO-FILLED gradient maps of those shapes (written through the const pointers), grad_losses (2, n_levels).
 * n_img <= 16. */
typedef struct rsdet_orpn_loss {
  int n_img, n_levels, A, num;
  int hw[8];
  const float* cls[8];
  const float* reg[8];
  const float* anchors;
  const int64_t* inside;
  const float* gt[16];
  int k_gt[16];
  const int64_t* inds;
  const uint8_t* is_pos;
  const uint8_t* val;
  const int64_t* assigned;
  const int64_t* counts;
  float means[6], stds[6];
  float beta, w_cls, w_box, pos_weight;
} rsdet_orpn_loss;
int rsdet_orpn_loss_rec_floats(int n_img, int num);
int rsdet_orpn_loss_forward_f32(const rsdet_orpn_loss* d, float* losses, float* rec, void* stream);
int rsdet_orpn_loss_backward_f32(const rsdet_orpn_loss* d, const float* rec, const float* grad_losses, void* stream);
/* MaxIoUAssigner on HORIZONTAL boxes without the (K, A) overlaps matrix.  Replaces models/boxes/assigner.py:65-170 with the
 * default BboxOverlaps2D calculator (models/boxes/iou_calculator.py:164-257, mode "iou", not aligned) as the Oriented RPN
 * calls it (models/roi_heads/oriented_rpn_head.py:292-300: 611 072 anchors per tile): rsdet_bbox_overlaps_f32 +
 * rsdet_assign_wrt_overlaps_f32 in two passes that recompute the IoU instead of storing it; gt_inds (and max_overlaps, NULL:
 * skipped) are bit-identical to that route.  gt (K, gt_stride >= 4), anchors (A, anchor_stride >= 4): (x1, y1, x2, y2, ...);
 * gt_inds (A) int32: assigned gt index + 1, 0 negative, -1 neither; K <= 1024, finite boxes, no ignore regions (the caller
 * keeps the matrix route for those).  ws: rsdet_hbb_assign_ws_size(K) bytes, 8-byte aligned. */
size_t rsdet_hbb_assign_ws_size(int K);
int rsdet_hbb_assign_f32(const float* gt, int K, int gt_stride, const float* anchors, int A, int anchor_stride, float eps,
                         float pos_iou_thr, float neg_lo, float neg_hi, float min_pos_iou, int match_low_quality,
                         int gt_max_assign_all, int32_t* gt_inds, float* max_overlaps, void* ws, size_t ws_bytes, void* stream);
/* OrientedHead's sampled RoIs and targets of ONE image, written into that image's rows of the batch.  Replaces
 * models/roi_heads/oriented_head.py:426-496 (get_bboxes_target_single) + arb2roi (:117-126) + the gathers of
 * SamplingResult (models/boxes/sampler.py:6-36) with OrientedDeltaXYWHTCoder.encode (models/boxes/coder.py:447-470), in the
 * fixed-size form of the train step.  inds / is_pos / val / assigned (num) from rsdet_sample_masked over the gt-extended
 * list: row i < k_gt is gt i, row k_gt + j is props[j] (rows of prop_stride >= 5 floats, obb first).  Outputs, num rows
 * each: rois (num, 6) = (image, obb), labels (int64: the gt's label for a positive, num_classes elsewhere), label_weights
 * (1 on used slots, pos_weight on positives when > 0, 0 on unused), bbox_targets / bbox_weights (num, 5). */
int rsdet_orcnn_roi_targets_f32(const float* props, int prop_stride, int n_props, const float* gt, const int64_t* gt_labels,
                                int k_gt, const int64_t* inds, const uint8_t* is_pos, const uint8_t* val,
                                const int64_t* assigned, int num, int image, int num_classes, const float* means,
                                const float* stds, float pos_weight, float* rois, int64_t* labels, float* label_weights,
                                float* bbox_targets, float* bbox_weights, void* stream);
/* The prepared weight operands of the backward (ops/weight_prep.py) refreshed by one launch: entries = n 64-byte records
 * in DEVICE memory {src, dst, var, gamma (pointers), O, C, T, eps (float), tile0, tiles_c, tiles_o, pad}: dst (C, T, O) =
 * src (O, T, C) with the taps reversed and row o scaled by gamma[o] / sqrt(var[o] + eps) (var NULL: copied).  tile0
 * ascending from 0, entry j owning T * tiles_c * tiles_o 32 x 32 tiles; total_tiles their sum. */
int rsdet_weight_prep_multi_bf16(const void* entries, int n, int total_tiles, void* stream);
int rsdet_conv3x3_fwd_mfma_bf16(const uint16_t* x, const uint16_t* weight, const float* bias, const uint8_t* live, int B,
                                int H, int W, int C, int O, int relu, uint16_t* out, void* stream);
/* Backward-data of the SECOND convolution of a conv + ReLU tower (s2anet_head.py:130-170: stacked ConvModules) through the
 * same kernel on the flipped weights, with the FIRST convolution's ReLU / bias backward in its epilogue:
 * grad_c1 = [c1 > 0] conv3x3(grad, weight_flipped), grad_bias1[o] = sum over positions of grad_c1 (NULL: not formed; ws of
 * rsdet_conv3x3_dgrad_gate_ws_size bytes when wanted).  grad (B,H,W,C), c1 / grad_c1 (B,H,W,O) bf16 channels-last. */
size_t rsdet_conv3x3_dgrad_gate_ws_size(int B, int H, int W, int O);
int rsdet_conv3x3_dgrad_gate_mfma_bf16(const uint16_t* grad, const uint16_t* weight_flipped, const uint16_t* c1, int B,
                                       int H, int W, int C, int O, uint16_t* grad_c1, float* grad_bias1, void* ws,
                                       size_t ws_bytes, void* stream);
/* Weight gradient of the same convolution (csrc/conv3x3_wrw_mfma.hip): split-K implicit GEMM over groups of image rows,
 * fragments by transposing LDS reads (both operands are position-major), fp32 partial tiles folded in a fixed order by a
 * second launch.  grad_out (B, H, W, O), x (B, H, W, C) channels-last bf16; grad_weight (O, 3, 3, C) bf16 (out_bf16 != 0)
 * or fp32; ws: rsdet_conv3x3_wrw_mfma_ws_size bytes, 16-byte aligned.  C % 64 == 0, O % 8 == 0. */
int rsdet_conv3x3_wrw_mfma_supported(int B, int H, int W, int C, int O);
size_t rsdet_conv3x3_wrw_mfma_ws_size(int B, int H, int W, int C, int O);
int rsdet_conv3x3_wrw_mfma_bf16(const uint16_t* grad_out, const uint16_t* x, int B, int H, int W, int C, int O,
                                void* grad_weight, int out_bf16, void* ws, size_t ws_bytes, void* stream);
/* ... of a convolution whose output feeds an eval-mode BatchNorm (conv2 / bn2 of the Bottleneck), from the gradient of the
 * BatchNorm's OUTPUT: row o times gamma[o] / sqrt(running_var[o] + eps) before the rounding, and rowdot[o] = sum of
 * weight[o] * (the unscaled fp32 row) for rsdet_bn_affine_grads_finish_multi_f32.  weight (O, 3, 3, C) bf16. */
int rsdet_conv3x3_wrw_mfma_rowscale_bf16(const uint16_t* grad_out, const uint16_t* x, int B, int H, int W, int C, int O,
                                         const float* running_var, const float* gamma, float eps, const uint16_t* weight,
                                         float* rowdot, void* grad_weight, int out_bf16, void* ws, size_t ws_bytes,
                                         void* stream);
int rsdet_alignconv_fwd_mfma_bf16(const uint16_t* im_nhwc, const float* offset, const uint16_t* weight,
                                  const rsdet_dcn_geom* g, int O, int out_nhwc, uint16_t* out, uint16_t* colT,
                                  void* stream);

/* ---- a18  ROIAlignRotated_v1 -----------------------------------------------------------------
 * Replaces _RotatedROIAlign_v1.execute / .grad: ops/roi_align_rotated_v1.py:300-351
 * (kernels :71-147, :193-298).  feat (N,C,H,W); rois (R,6) = (batch, cx, cy, w, h, theta);
 * out / grad_out (R,C,PH,PW).  backward ACCUMULATES into grad_feat (zero it first, :345). */
int rsdet_rroi_align_v1_forward_f32(const float* feat, const float* rois, int R, int C, int H,
                                    int W, int PH, int PW, float spatial_scale, int sample_num,
                                    float* out, void* stream);
/* The forward of OrientedSingleRoIExtractor.execute (models/roi_extractors/oriented_single_level.py:91-114: map_roi_levels,
 * then one ROIAlignRotated per level on that level's RoIs) in one launch: RoI n samples feat[lvl[n]] (N,C,H_l,W_l) at
 * scale[l].  lvl (R,) int32, clamped to [0, n_levels).  Same arithmetic per RoI as the single-map entry points. */
#define RSDET_RROI_MAX_LEVELS 8
typedef struct rsdet_rroi_levels {
  int n_levels;
  const float* feat[RSDET_RROI_MAX_LEVELS];
  int H[RSDET_RROI_MAX_LEVELS], W[RSDET_RROI_MAX_LEVELS];
  float scale[RSDET_RROI_MAX_LEVELS];
} rsdet_rroi_levels;
int rsdet_rroi_align_v1_forward_levels_f32(const rsdet_rroi_levels* levels, const float* rois, const int32_t* lvl, int R,
                                           int C, int PH, int PW, int sample_num, float* out, void* stream);
int rsdet_rroi_align_v0_forward_levels_f32(const rsdet_rroi_levels* levels, const float* rois, const int32_t* lvl, int R,
                                           int C, int PH, int PW, int sample_num, float* out, void* stream);
/* Its backward (the .grad of the per-level ROIAlignRotated calls, roi_align_rotated_v1.py:329-351, for all levels): ONE
 * inverted index over the levels' pixels laid end to end (every RoI on the geometry of its own level), then a gather per
 * level straight into grad_feat[l] (N,C,H_l,W_l) -- written completely, no zero fill; NULL = level not wanted.
 * grad_out_t (R, PH*PW, C): the output gradient with channels last; C % 4 == 0; sample_num >= 1; feat[] unused. */
size_t rsdet_rroi_align_backward_levels_ws_size(const rsdet_rroi_levels* levels, int R, int PH, int PW, int sample_num, int N);
int rsdet_rroi_align_v1_backward_levels_nchw_f32(const rsdet_rroi_levels* levels, float* const* grad_feat,
                                                 const float* grad_out_t, const float* rois, const int32_t* lvl, int R,
                                                 int C, int N, int PH, int PW, int sample_num, void* ws, size_t ws_bytes,
                                                 void* stream);
int rsdet_rroi_align_v0_backward_levels_nchw_f32(const rsdet_rroi_levels* levels, float* const* grad_feat,
                                                 const float* grad_out_t, const float* rois, const int32_t* lvl, int R,
                                                 int C, int N, int PH, int PW, int sample_num, void* ws, size_t ws_bytes,
                                                 void* stream);
int rsdet_rroi_align_v1_backward_f32(const float* grad_out, const float* rois, int R, int C,
                                     int H, int W, int PH, int PW, float spatial_scale,
                                     int sample_num, float* grad_feat, void* stream);

/* Gather form of the backward for a fixed sample_num (> 0): grad_out_t is the output gradient in channels-last
 * form (R, PH*PW, C), grad_feat_nhwc (N, H, W, C) is written completely (no zero fill, no fp32 atomics); the caller
 * converts layouts.  (roi, bin, sample, corner) -> pixel is inverted on integers in ws first. */
size_t rsdet_rroi_align_v1_backward_gather_ws_size(int R, int PH, int PW, int sample_num, int N, int H, int W);
int rsdet_rroi_align_v1_backward_gather_f32(const float* grad_out_t, const float* rois, int R, int C, int N, int H,
                                            int W, int PH, int PW, float spatial_scale, int sample_num,
                                            float* grad_feat_nhwc, void* ws, size_t ws_bytes, void* stream);
/* ... and with the result written in NCHW (N, C, H, W), the layout of the features and of the convolution that
 * receives the gradient: every element written exactly once, no (N,H,W,C) intermediate, no transposes of the
 * feature-sized gradient (workgroup = 64 pixels, 64-channel chunks transposed through LDS). */
int rsdet_rroi_align_v1_backward_gather_nchw_f32(const float* grad_out_t, const float* rois, int R, int C, int N,
                                                 int H, int W, int PH, int PW, float spatial_scale, int sample_num,
                                                 float* grad_feat_nchw, void* ws, size_t ws_bytes, void* stream);

/* ---- f4  ROIAlignRotated (v0) -----------------------------------------------------------------
 * Replaces _RotatedROIAlign.execute / .grad: ops/roi_align_rotated.py:256-309 (kernels :59-126, :170-254).
 * Same tensors and calling rules as the v1 entries above; differs in the RoI frame only (no -0.5 pixel shift of
 * the centre :76-77, opposite rotation sense :116-117).  The gather form uses
 * rsdet_rroi_align_v1_backward_gather_ws_size for its workspace. */
/* The gather-form backward in two calls, so that the inverted index -- a function of the RoIs and the geometry, not of the
 * gradient -- can be built at FORWARD time beside the forward kernel: _backward_index_f32 fills `ws` (sized by
 * rsdet_rroi_align_v1_backward_gather_ws_size; count + scan + fill), rsdet_rroi_align_backward_gather_indexed_f32 is the
 * gather alone on that workspace (nchw != 0: grad_feat (N, C, H, W), else (N, H, W, C)); the pair computes what the one-call
 * forms above compute (same entries and weights). */
int rsdet_rroi_align_v1_backward_index_f32(const float* rois, int R, int N, int H, int W, int PH, int PW,
                                           float spatial_scale, int sample_num, void* ws, size_t ws_bytes, void* stream);
int rsdet_rroi_align_v0_backward_index_f32(const float* rois, int R, int N, int H, int W, int PH, int PW,
                                           float spatial_scale, int sample_num, void* ws, size_t ws_bytes, void* stream);
int rsdet_rroi_align_backward_gather_indexed_f32(const float* grad_out_t, int R, int C, int N, int H, int W, int PH, int PW,
                                                 int sample_num, int nchw, float* grad_feat, const void* ws, size_t ws_bytes,
                                                 void* stream);
int rsdet_rroi_align_v0_forward_f32(const float* feat, const float* rois, int R, int C, int H,
                                    int W, int PH, int PW, float spatial_scale, int sample_num,
                                    float* out, void* stream);
int rsdet_rroi_align_v0_backward_f32(const float* grad_out, const float* rois, int R, int C,
                                     int H, int W, int PH, int PW, float spatial_scale,
                                     int sample_num, float* grad_feat, void* stream);
int rsdet_rroi_align_v0_backward_gather_f32(const float* grad_out_t, const float* rois, int R, int C, int N, int H,
                                            int W, int PH, int PW, float spatial_scale, int sample_num,
                                            float* grad_feat_nhwc, void* ws, size_t ws_bytes, void* stream);
int rsdet_rroi_align_v0_backward_gather_nchw_f32(const float* grad_out_t, const float* rois, int R, int C, int N,
                                                 int H, int W, int PH, int PW, float spatial_scale, int sample_num,
                                                 float* grad_feat_nchw, void* ws, size_t ws_bytes, void* stream);

/* ---- f4  FeatureRefine (R3Det) --------------------------------------------------------------------
 * Replaces feature_refine_forward / feature_refine_backward: ops/fr.py:234-252 (kernels :113-173, :175-232).
 * feat / out (N,C,H,W); best_bboxes (N,H,W,5); points in {1, 5} (:261).  out = feat + sum over the points of the
 * bilinear sample of the same channel; box entry 0 is the ROW coordinate, entry 1 the COLUMN (:131-133, kept).
 * Backward takes and returns CHANNELS-LAST gradients (N,H,W,C) (the caller converts): the point -> pixel map is
 * inverted on integers in ws, then every pixel sums its terms -- no fp32 atomics, grad_in written exactly once. */
int rsdet_feature_refine_forward_f32(const float* feat, const float* best_bboxes, int N, int C, int H, int W,
                                     float spatial_scale, int points, float* out, void* stream);
/* The forward on CHANNELS-LAST maps feat / out (N,H,W,C) (a channels_last step hands them over as they are; pairs with
 * the channels-last backward below, so neither direction turns a layout): a sample is a contiguous channel vector, read
 * by 16-byte-per-lane wave-wide loads.  Same sums in the same order as the NCHW entry.  C % 4 == 0 with C / 4 a power of
 * two up to 64 or a multiple of 64 (rsdet_feature_refine_forward_nhwc_supported), H W C < 2^31, 16-byte aligned maps. */
int rsdet_feature_refine_forward_nhwc_supported(int C);
int rsdet_feature_refine_forward_nhwc_f32(const float* feat, const float* best_bboxes, int N, int C, int H, int W,
                                          float spatial_scale, int points, float* out, void* stream);
size_t rsdet_feature_refine_backward_ws_size(int N, int H, int W, int points);
int rsdet_feature_refine_backward_nhwc_f32(const float* grad_out_nhwc, const float* best_bboxes, int N, int C,
                                           int H, int W, float spatial_scale, int points, float* grad_in_nhwc,
                                           void* ws, size_t ws_bytes, void* stream);

/* ---- f4  convex_sort ---------------------------------------------------------------------------------
 * Replaces convex_sort (ops/convex_sort.py:67-201; caller models/losses/poly_iou_loss.py:23): for each of nbs
 * point sets pts (nbs, npts, 2) with masks (nbs, npts) (a point takes part iff mask >= 0.5) the indices of the
 * convex hull in scan order, starting at the lowest valid point, closed by the start index when `circular`;
 * convex_index (nbs, circular ? npts + 1 : npts) int32, unused slots -1.  The masked argmin, the cosine keys, the
 * descending sort (:159-176) and the Graham scan (:5-64) run in one launch.  Tie rules (Jittor's are unpinned):
 * first index for the argmin, stable order for equal keys.  ws: rsdet_convex_sort_ws_size bytes (0 for npts <= 56). */
size_t rsdet_convex_sort_ws_size(int nbs, int npts);
int rsdet_convex_sort_f32(const float* pts, const float* masks, int nbs, int npts, int circular, int* convex_index,
                          void* ws, size_t ws_bytes, void* stream);

/* ---- f4  poly_nms (in-model polygon NMS, fp32) -------------------------------------------------------
 * Replaces poly_nms: ops/nms_poly.py:186-210 (mask kernel :135-183, quadrilateral IoU devPolyIoU :17-132, host
 * sweep :195-207); caller multiclass_poly_nms :212-224 <- roi_heads/gliding_head.py:181.
 * dets_sorted (n, 9) = x1,y1,...,x4,y4,score in descending-score order; keep_sorted[i] = 1 iff kept; box i
 * suppresses a later box j when IoU(i, j) > thr (:179).  The IoU is the reference's float arithmetic restated
 * operation by operation (its cancellation noise decides borderline pairs, so nothing is gated or reordered).
 * ws: rsdet_nms_hbb_ws_size(n) bytes.  rsdet_poly_iou_f32: the same IoU as a dense (n1, n2) matrix over (n, 8)
 * quadrilaterals. */
int rsdet_poly_nms_sorted_f32(const float* dets_sorted, int n, float thr, uint8_t* keep_sorted, void* ws,
                              size_t ws_bytes, void* stream);
int rsdet_poly_iou_f32(const float* polys1, int n1, const float* polys2, int n2, float* ious, void* stream);

/* ---- a17  rotated_box_to_poly ------------------------------------------------------------------
 * Replaces models/boxes/box_ops.py:633-654.  boxes (n,5) -> polys (n,8). */
int rsdet_rotated_box_to_poly_f32(const float* boxes, int n, float* polys, void* stream);

/* ---- a20  depthwise convolutions of the VAN backbone ---------------------------------------------------
 * Replaces nn.Conv2d(dim, dim, k, groups=dim) at models/backbones/van.py:32 (3x3), :56 (5x5), :57 (7x7, dilation 3)
 * on the Oriented R-CNN + VAN path: stride 1, "same" padding dilation*(K-1)/2, (K, dilation) in {(3,1), (5,1), (7,3)}
 * (anything else: RSDET_EINVAL -- the host module then keeps torch's convolution).  x / y / grad (N,C,H,W),
 * weight (C,1,K,K) = (C,K,K) contiguous, bias (C) or NULL.
 * in_bias (C) or NULL: a per-channel constant added to x before the convolution (zero padding stays zero) -- the bias
 * of the 1x1 convolution in front of the depthwise one in the VAN MLP (van.py:43-51: fc1 -> dwconv), which then runs
 * without its bias kernel; backward_data returns that bias' gradient = sum(grad_x) in grad_in_bias (NULL: not wanted;
 * ws of rsdet_dwconv2d_backward_data_ws_size bytes only when it is).  backward_weight takes the same in_bias, also
 * returns the bias gradient of the depthwise convolution itself (grad_bias may be NULL).  All sums run in a fixed order
 * (two stages through ws, no float atomics).   ws WITHOUT grad_in_bias: the per-tile sums stay in ws as [c][slot] floats (ws_size / 4 / C
 * slots per channel) for a consumer that folds them itself (rsdet_van_fold_bn_f32's gs_tab). */
int rsdet_dwconv2d_forward_f32(const float* x, const float* in_bias, const float* weight, const float* bias, int N,
                               int C, int H, int W, int K, int dilation, float* y, void* stream);
size_t rsdet_dwconv2d_backward_data_ws_size(int N, int C, int H, int W);
int rsdet_dwconv2d_backward_data_f32(const float* grad_y, const float* weight, int N, int C, int H, int W, int K,
                                     int dilation, float* grad_x, float* grad_in_bias, void* ws, size_t ws_bytes,
                                     void* stream);
size_t rsdet_dwconv2d_backward_weight_ws_size(int N, int C, int H, int W, int K);
/* rsdet_dwconv2d_forward_f32 with a second output y_act = GELU(y) (erf form): Mlp.dwconv + Mlp.act as one pass
 * (van.py:169-171); y_is_grad != 0: y receives GELU'(conv) instead of the convolution's value -- all the backward of the
 * activation needs (the backward-data GEMM of fc2 multiplies by it in its epilogue); rsdet_dwconv2d_backward_data_f32 whose result meets a second gradient of the same tensor and a GELU:
 * grad_x = (dwconv^T(grad_y) + add) * GELU'(gelu_arg), grad_sum[c] = sum of grad_x over the map (van.py:177-215: u =
 * GELU(proj_1(x)) is read by conv0 AND by the gate). */
int rsdet_dwconv2d_forward_act_f32(const float* x, const float* weight, const float* bias, int N, int C, int H, int W, int K,
                                   int dilation, int y_is_grad, float* y, float* y_act, void* stream);
int rsdet_dwconv2d_backward_data_act_f32(const float* grad_y, const float* weight, int N, int C, int H, int W, int K,
                                         int dilation, const float* add, const float* gelu_arg, float* grad_x,
                                         float* grad_sum, void* ws, size_t ws_bytes, void* stream);
int rsdet_dwconv2d_backward_weight_f32(const float* grad_y, const float* x, const float* in_bias, int N, int C, int H,
                                       int W, int K, int dilation, float* grad_weight, float* grad_bias, void* ws,
                                       size_t ws_bytes, void* stream);

/* ---- a21  eval-mode BatchNorm + residual add + ReLU (backbone Bottleneck tails) -----------------------
 * Replaces the per-op sequence models/backbones/resnet.py:101-126 (bn -> (+ identity) -> relu) when the
 * BatchNorm is in eval mode (norm_eval, :177-184): y = max(((x - mean) * rsqrt(var + eps)) * weight + bias
 * (+ residual), 0) on NCHW fp32, one HBM pass.  weight / bias / residual may be NULL; relu 0 skips the max.
 * Backward: g = grad_y * [y > 0] (relu) -> grad_residual = g, grad_x = g * rsqrt(var+eps) * weight,
 * grad_weight[c] = sum g * xhat, grad_bias[c] = sum g (two-stage, fixed order: deterministic).  Any of the
 * four outputs may be NULL; ws is only needed for grad_weight / grad_bias. */
size_t rsdet_bn_act_backward_ws_size(int N, int C, int HW);
int rsdet_bn_act_forward_f32(const float* x, const float* residual, const float* running_mean,
                             const float* running_var, const float* weight, const float* bias, float eps, int N,
                             int C, int HW, int relu, float* y, void* stream);
int rsdet_bn_act_backward_f32(const float* grad_y, const float* y, const float* x, const float* running_mean,
                              const float* running_var, const float* weight, float eps, int N, int C, int HW,
                              int relu, float* grad_x, float* grad_residual, float* grad_weight, float* grad_bias,
                              void* ws, size_t ws_bytes, void* stream);
/* The same two passes for bf16 activations (the autocast step of BASELINE configs 2 and 4; same reference lines):
 * x / residual / y / grad_* tensors are bf16 (16-bit storage), the arithmetic, the parameters and the parameter
 * gradients fp32; stores round to nearest even. */
int rsdet_bn_act_forward_bf16(const uint16_t* x, const uint16_t* residual, const float* running_mean,
                              const float* running_var, const float* weight, const float* bias, float eps, int N,
                              int C, int HW, int relu, uint16_t* y, void* stream);
int rsdet_bn_act_backward_bf16(const uint16_t* grad_y, const uint16_t* y, const uint16_t* x,
                               const float* running_mean, const float* running_var, const float* weight, float eps,
                               int N, int C, int HW, int relu, uint16_t* grad_x, uint16_t* grad_residual,
                               float* grad_weight, float* grad_bias, void* ws, size_t ws_bytes, void* stream);

/* in (B, R, C) -> out (B, C, R), fp32, through 64 x 64 LDS tiles: the NCHW <-> NHWC turns around the channels-last
 * gather kernels (ops/roi_align_rotated_v1.py:329-351, ops/fr.py:235-260 and ops/dcn_v1.py:456-557 consume and produce
 * NCHW).  NHWC -> NCHW: B = N, R = H*W, C = channels.  in != out; R <= 64*65535. */
int rsdet_transpose_last2_f32(const float* in, float* out, int B, int R, int C, void* stream);

/* Channels-last convolution weights for "backward-data as a forward convolution" (a stride-1 / same-padding convolution's
 * input gradient is the forward convolution of the output gradient with the taps reversed and the channel axes
 * exchanged; the reference's convolutions: models/utils/modules.py ConvModule, models/backbones/resnet.py:101-126):
 * in (O, T, C) -> out (C, T, O) with tap t -> T - 1 - t; elem_bytes 2 or 4, bits copied. */
int rsdet_weight_flip_transpose(const void* in, void* out, int O, int C, int T, int elem_bytes, void* stream);

/* Column sums of a (rows, C) matrix, C <= 64: out[c] = sum_r x[r, c] in fp32 (two deterministic stages).  The bias
 * gradient of a channels_last convolution with few output channels (the 5- / 15-channel prediction maps of
 * models/roi_heads/s2anet_head.py:128-142): rows = N*H*W. */
size_t rsdet_colsum_ws_size(long long rows, int C);
int rsdet_colsum_f32(const float* x, long long rows, int C, float* out, void* ws, size_t ws_bytes, void* stream);
int rsdet_colsum_bf16(const uint16_t* x, long long rows, int C, float* out, void* ws, size_t ws_bytes, void* stream);

/* Channels-last (NHWC) forms of the same four entries, for the bf16 trunk that runs channels_last (MIOpen's bf16
 * convolutions are NHWC-native; on NCHW tensors every one of them is wrapped in layout transposes).  x / y / residual /
 * gradients are (N, H, W, C) contiguous, i.e. torch tensors of shape (N, C, H, W) in channels_last memory format; same
 * arithmetic, same deterministic two-stage parameter gradients.  C must satisfy rsdet_bn_act_nhwc_supported(C)
 * (C % 4 == 0 and C / 4 a divisor or a multiple (<= 4x) of 256: every ResNet / FPN width). */
/* Stem tail (models/backbones/resnet.py:186-189 of the reference: bn1 -> relu -> maxpool 3x3 / stride 2 / padding 1) as one
 * forward pass over a channels-last map: y (N, (H+1)/2, (W+1)/2, C) <- maxpool(relu(bn(x))), x (N, H, W, C).  bf16 != 0:
 * bfloat16 elements and C % 8 == 0, else float and C % 4 == 0.  Forward only (the shipped configs freeze the stem). */
int rsdet_bn_relu_maxpool_nhwc(const void* x, int bf16, const float* running_mean, const float* running_var,
                               const float* weight, const float* bias, float eps, int N, int C, int H, int W, void* y,
                               void* stream);
int rsdet_bn_act_nhwc_supported(int C);
size_t rsdet_bn_act_backward_nhwc_ws_size(int N, int C, int HW);
int rsdet_bn_act_forward_nhwc_f32(const float* x, const float* residual, const float* running_mean,
                                  const float* running_var, const float* weight, const float* bias, float eps,
                                  int N, int C, int HW, int relu, float* y, void* stream);
int rsdet_bn_act_backward_nhwc_f32(const float* grad_y, const float* y, const float* x, const float* running_mean,
                                   const float* running_var, const float* weight, float eps, int N, int C, int HW,
                                   int relu, float* grad_x, float* grad_residual, float* grad_weight,
                                   float* grad_bias, void* ws, size_t ws_bytes, void* stream);
int rsdet_bn_act_forward_nhwc_bf16(const uint16_t* x, const uint16_t* residual, const float* running_mean,
                                   const float* running_var, const float* weight, const float* bias, float eps,
                                   int N, int C, int HW, int relu, uint16_t* y, void* stream);
int rsdet_bn_act_backward_nhwc_bf16(const uint16_t* grad_y, const uint16_t* y, const uint16_t* x,
                                    const float* running_mean, const float* running_var, const float* weight,
                                    float eps, int N, int C, int HW, int relu, uint16_t* grad_x,
                                    uint16_t* grad_residual, float* grad_weight, float* grad_bias, void* ws,
                                    size_t ws_bytes, void* stream);

/* The same pair with the ReLU gate carried as ONE BIT per element instead of re-read from y: the forward writes
 * rsdet_bn_act_relu_mask_bytes(N, C, HW, bf16) bytes (0 = this shape has no mask form: use the entries above), the
 * backward reads them in place of y -- 19 % less HBM traffic for the backward (resnet.py:101-126 is the fused sequence;
 * nothing in the reference corresponds to the mask, its autograd keeps y).  relu is implied (= 1). */
size_t rsdet_bn_act_relu_mask_bytes(int N, int C, int HW, int bf16);
int rsdet_bn_act_forward_nhwc_mask_f32(const float* x, const float* residual, const float* running_mean,
                                       const float* running_var, const float* weight, const float* bias, float eps,
                                       int N, int C, int HW, int relu, float* y, uint8_t* relu_mask, void* stream);
int rsdet_bn_act_forward_nhwc_mask_bf16(const uint16_t* x, const uint16_t* residual, const float* running_mean,
                                        const float* running_var, const float* weight, const float* bias, float eps,
                                        int N, int C, int HW, int relu, uint16_t* y, uint8_t* relu_mask, void* stream);
int rsdet_bn_act_backward_nhwc_mask_f32(const float* grad_y, const uint8_t* relu_mask, const float* x,
                                        const float* running_mean, const float* running_var, const float* weight,
                                        float eps, int N, int C, int HW, float* grad_x, float* grad_residual,
                                        float* grad_weight, float* grad_bias, void* ws, size_t ws_bytes, void* stream);
/* ... for an output that was used twice (the output of a residual block feeds the next block's first convolution AND its
 * identity branch, models/backbones/resnet.py:80-91): the two gradients are summed while they are read, instead of by a pass
 * of the autograd engine's own over the trunk's widest tensors. */
int rsdet_bn_act_backward_nhwc_mask2_f32(const float* grad_y, const float* grad_y2, const uint8_t* relu_mask, const float* x,
                                         const float* running_mean, const float* running_var, const float* weight, float eps,
                                         int N, int C, int HW, float* grad_x, float* grad_residual, float* grad_weight,
                                         float* grad_bias, void* ws, size_t ws_bytes, void* stream);
int rsdet_bn_act_backward_nhwc_mask_bf16(const uint16_t* grad_y, const uint8_t* relu_mask, const uint16_t* x,
                                         const float* running_mean, const float* running_var, const float* weight,
                                         float eps, int N, int C, int HW, uint16_t* grad_x, uint16_t* grad_residual,
                                         float* grad_weight, float* grad_bias, void* ws, size_t ws_bytes, void* stream);

/* ---- 8(f) rank 1  polygon IoU + tile-merge polygon NMS (evaluation side) ------------------------------------
 * Replaces ops/nms_poly.py:247-252 (iou_poly: shapely intersection area, max(union, 0.01) in the denominator),
 * data/devkits/result_merge.py:66-126 (py_cpu_nms_poly_fast) and the overlap loop of
 * data/devkits/voc_eval.py:263-304.  Quadrilaterals are 8 doubles (x1,y1,...,x4,y4), either orientation; at least
 * one polygon of a pair must be convex (detections are rectangles).  shapely is not available: parity unpinned.
 * rsdet_nms_poly_sorted_f64: polys already in descending-score order; keep_sorted[i] = 1 if kept; horizontal-hull
 * gate and `IoU > thr` suppression exactly as py_cpu_nms_poly_fast; ws sized by rsdet_nms_hbb_ws_size(n). */
int rsdet_poly_iou_f64(const double* polys1, int n1, const double* polys2, int n2, double* ious, void* stream);
int rsdet_nms_poly_sorted_f64(const double* polys_sorted, int n, double thr, uint8_t* keep_sorted, void* ws,
                              size_t ws_bytes, void* stream);

/* ---- optimizer step over all parameters in two launches (csrc/optim.hip) ---------------------------------------------
 * Replaces optims/optimizer.py:24-43 (grad_clip max_norm / norm_type 2, then SGD with weight decay and momentum) and,
 * for bf16 model weights, the per-parameter master <-> bf16 casts of the autocast route.
 *   tensors: n_tensors 64-byte device records { const void* grad; void* param; float* master; float* mom;
 *            long long n; int flags; pad; float* mom2; pad } -- flags bit 0: grad is bf16, bit 1: param is bf16 (then master != NULL holds
 *            the fp32 parameter; otherwise param is the fp32 parameter itself);
 *   chunks:  n_chunks device int pairs { tensor index, chunk index } covering every tensor in pieces of
 *            rsdet_mt_chunk_elems() elements;
 *   max_norm <= 0: no clipping (launch 1 is skipped unless sqnorm_out is given); sqnorm_out: optional device float that
 *            receives the squared global gradient norm;
 *   state:   rsdet_mt_sgd_state_bytes(n_chunks) bytes, zero on entry, left zero.
 * m = momentum * m + (grad * clip_coef + weight_decay * p);  p -= lr * m   (torch.optim.SGD, dampening 0). */
int rsdet_mt_chunk_elems(void);
size_t rsdet_mt_sgd_state_bytes(int n_chunks);
int rsdet_mt_sgd_step(const void* tensors, const int* chunks, int n_chunks, float max_norm, float lr, float momentum,
                      float weight_decay, float* sqnorm_out, void* state, size_t state_bytes, void* stream);
/* AdamW over the same records and chunks (optims/optimizer.py:24-43 with jittor.optim.AdamW, the optimizer of
 * configs/orcnn/orcnn_van3_7_anchor.py): the record's 8 bytes after `flags; pad` hold `float* mom2` (exp_avg_sq), `mom`
 * is exp_avg.  step = 1-based count of this update.  p *= 1 - lr * wd;  m += (g - m)(1 - beta1);
 * v = v * beta2 + (1 - beta2) g^2;  p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps). */
int rsdet_mt_adamw_step(const void* tensors, const int* chunks, int n_chunks, float max_norm, double lr, double beta1,
                        double beta2, double eps, double weight_decay, long long step, float* sqnorm_out, void* state,
                        size_t state_bytes, void* stream);

/* ---- the elementwise tails of a VAN block, fused (csrc/van_ops.hip; NCHW fp32) ------------------------------------------
 * Replace, with the 1x1 convolutions run without their bias, the bias adds, GELU, gate product, shortcut and layer-scale
 * passes of /root/reference/python/jdet/models/backbones/van.py:46-122 (Mlp, LKA / Attention, Block.execute) and their
 * backward passes incl. the per-channel bias / scale reductions (deterministic two-stage sums).  Maps are (N, C, HW)
 * contiguous; bias / shortcut may be NULL; gbias / gscale NULL = not wanted; ws: rsdet_van_ws_size(N, C, HW) bytes.
 *   bias_gelu  y = GELU(x + b[c]) (erf form)         bwd: gx = gy * GELU'(x + b), gbias = sum gx
 *   gate       y = u * (a + b[c])                    bwd: gu = g * (a + b), ga = g * u, gbias = sum ga
 *   residual   y = x + scale[c] * (p + b[c] + shortcut)
 *              bwd: gp = scale * g (= the shortcut's gradient), gbias = scale * sum g, gscale = sum g * (p + b + shortcut);
 *              the gradient of x is g itself. */
int rsdet_van_supported(int N, int C, int HW);
size_t rsdet_van_ws_size(int N, int C, int HW);
int rsdet_van_bias_gelu_fwd_f32(const float* x, const float* bias, int N, int C, int HW, float* y, void* stream);
int rsdet_van_bias_gelu_bwd_f32(const float* gy, const float* x, const float* bias, int N, int C, int HW, float* gx,
                                float* gbias, void* ws, size_t ws_bytes, void* stream);
int rsdet_van_gate_fwd_f32(const float* u, const float* a, const float* bias, int N, int C, int HW, float* y,
                           void* stream);
int rsdet_van_gate_bwd_f32(const float* g, const float* u, const float* a, const float* bias, int N, int C, int HW,
                           float* gu, float* ga, float* gbias, void* ws, size_t ws_bytes, void* stream);
int rsdet_van_residual_fwd_f32(const float* x, const float* p, const float* bias, const float* shortcut,
                               const float* scale, int N, int C, int HW, float* y, void* stream);
int rsdet_van_residual_bwd_f32(const float* g, const float* p, const float* bias, const float* shortcut,
                               const float* scale, int N, int C, int HW, float* gp, float* gbias, float* gscale, void* ws,
                               size_t ws_bytes, void* stream);

/* ---- the pyramid canvas of the S2ANet head (csrc/canvas.hip) ----------------------------------------------------------
 * The reference applies the head's shared-weight convolutions level by level
 * (/root/reference/python/jdet/models/roi_heads/s2anet_head.py:207-255).  These entry points lay the L <= 8 level maps of
 * a batch side by side in one (B, C, Hc, Wc) canvas with zero gaps, so that each convolution runs once:
 *   pixmap:  canvas_pixels device ints, -1 for a gap pixel, else (level << 27) | pixel index inside that level;
 *   rsdet_pyramid_copy: to_canvas != 0: canvas <- levels, gap pixels <- 0; to_canvas == 0: levels <- canvas.  Either side
 *            NCHW (flag 0) or channels-last (flag 1); elem_bytes 2 or 4 (bits are copied, no conversion);
 *   rsdet_canvas_bias_act_*: y = act(x + bias[c]) at live pixels (live[p] != 0), 0 at gap pixels; act = ReLU (relu != 0)
 *            or identity.  nhwc: C % 4 == 0.  Backward of the ReLU form = rsdet_bn_act_backward_* gated on y > 0. */
int rsdet_pyramid_copy(void* const* levels, const int* level_pixels, int n_levels, void* canvas, const int* pixmap, int B,
                       int C, int canvas_pixels, int elem_bytes, int canvas_nhwc, int levels_nhwc, int to_canvas,
                       void* stream);
int rsdet_canvas_bias_act_f32(const float* x, const float* bias, const uint8_t* live, int N, int C, int HW, int relu,
                              int nhwc, float* y, void* stream);
int rsdet_canvas_bias_act_bf16(const uint16_t* x, const float* bias, const uint8_t* live, int N, int C, int HW, int relu,
                               int nhwc, uint16_t* y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RSDET_H_ */
