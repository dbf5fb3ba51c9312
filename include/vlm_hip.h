/* vlm_hip.h -- C ABI of libvlm_hip.so: the MI355X (gfx950) hot path of ylsung/vl-merging.
 *
 * The reference (/root/reference) is 100 % Python and has no native boundary of its own; every entry
 * point below replaces a stock-PyTorch op site of the reference's hot path and cites it (file:line under
 * /root/reference/src).  Conventions:
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless named *_host;
 *   - every call is stream-ordered on `stream` (a hipStream_t passed as void*), re-entrant, allocates
 *     nothing: the caller owns outputs and workspaces;
 *   - return 0 on success, a negative VLM_ERR_* otherwise; nothing throws across the boundary.
 *
 * Token layout ("segment-major"): a pass over B samples with n0 text and n1 image tokens per sample keeps
 * activations as a [rows, D] matrix whose rows are  base0 + b*n0 + t  (t < n0)  and  base1 + b*n1 + (t-n0).
 * Modality-specific experts (all_moe) then see contiguous row ranges (vision_transformer.py:607-681 slices
 * and torch.cat's per layer instead).
 */
#ifndef VLM_HIP_H
#define VLM_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VLM_OK 0
#define VLM_ERR_ARG (-1)
#define VLM_ERR_LAUNCH (-2)
#define VLM_ERR_WORKSPACE (-3)
#define VLM_ERR_UNSUPPORTED (-4)

#define VLM_ABI_VERSION 1
int vlm_abi_version(void);
/* Number of compute units of the current device (grid sizing), or negative error. */
int vlm_device_cus(void);

/* ------------------------------------------------------------------------------------------------
 * Checkpoint merge (K12/K13/K14-bias): modules/vilt_module.py:533-638 (merge_weights),
 * :640-746 (sum_task_vectors), :436-457/:486-529 (regmean's plain averages).
 * One job = one output tensor.  Arithmetic is fp32, unfused (no FMA), in source order:
 *   VLM_MERGE_LERP    : acc = 0;        acc = acc + ratio[m] * src[m]          (:590-601)
 *   VLM_MERGE_TASKVEC : acc = base;     acc = acc + ratio[m] * (src[m] - acc)  (:700-706, in-place alias
 *                       of the central tensor => recurrence, SURVEY.md 8a checklist item 8)
 *   VLM_MERGE_MEAN    : acc = 0;        acc = acc + src[m];  acc / n_src       (:436-457)
 * dst may alias base.  Bit-exact with the reference's CPU path.
 */
#define VLM_MERGE_LERP 0
#define VLM_MERGE_TASKVEC 1
#define VLM_MERGE_MEAN 2
#define VLM_MERGE_MAX_SRC 4

typedef struct {
  void* dst;                            /* f32 [n_elem] */
  const void* base;                     /* f32 [n_elem], TASKVEC only (central weight) */
  const void* src[VLM_MERGE_MAX_SRC];   /* f32 [n_elem] each */
  float ratio[VLM_MERGE_MAX_SRC];
  int32_t n_src;
  int32_t mode;
  uint64_t n_elem;
} vlm_merge_job_t;

/* Bytes of device workspace a plan for n_jobs jobs over total_elems elements needs. */
size_t vlm_merge_plan_bytes(int n_jobs, uint64_t total_elems);
/* Build the chunk table on the host and copy jobs + table into `workspace` (stream-ordered H2D from a
 * pageable host buffer: the call returns after the copy is enqueued and the source is no longer needed). */
int vlm_merge_plan_upload(const vlm_merge_job_t* jobs_host, int n_jobs, void* workspace, size_t workspace_bytes,
                          void* stream);
/* Run an uploaded plan: ONE kernel launch over all jobs. */
int vlm_merge_run(const void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VLM_HIP_H */
