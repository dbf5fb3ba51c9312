/* CPU restatement of the reference's merge arithmetic -- TEST INFRASTRUCTURE ONLY (the oracle).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Follows /root/reference/src/vilt/modules/vilt_module.py:
 *   merge_lerp    :590-601 / :607-618 / :624-635   later_weight = 0; later_weight += ratio[m] * W_m
 *   merge_taskvec :700-706 / :717-724 / :735-742   c = central (alias); c += lam * (W_m - c)   (in place)
 *   merge_mean    :436-457 / :486-529              s = 0; s += W_m; s / count
 * PyTorch's CPU kernels multiply an fp32 tensor by fp32(ratio), round, then add and round again (no FMA):
 * build with -ffp-contract=off.  Pinned against tests/golden/merge_tiny.npz and merge_base_digests.json,
 * both produced by the reference itself (tests/golden/make_golden.py).
 */
#include <stddef.h>
#include <stdint.h>

void vlm_ref_merge_lerp(float* dst, const float* const* src, const float* ratio, int n_src, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    float acc = 0.0f; /* int 0 + tensor => (+0.0) + x : normalises -0.0 like the reference */
    for (int m = 0; m < n_src; ++m) {
      float t = ratio[m] * src[m][i];
      acc = acc + t;
    }
    dst[i] = acc;
  }
}

void vlm_ref_merge_taskvec(float* dst, const float* base, const float* const* src, const float* ratio, int n_src,
                           size_t n) {
  for (size_t i = 0; i < n; ++i) {
    float c = base[i];
    for (int m = 0; m < n_src; ++m) {
      float d = src[m][i] - c;
      float t = ratio[m] * d;
      c = c + t;
    }
    dst[i] = c;
  }
}

void vlm_ref_merge_mean(float* dst, const float* const* src, int n_src, size_t n) {
  const float cnt = (float)n_src;
  for (size_t i = 0; i < n; ++i) {
    float acc = 0.0f;
    for (int m = 0; m < n_src; ++m) acc = acc + src[m][i];
    dst[i] = acc / cnt;
  }
}
