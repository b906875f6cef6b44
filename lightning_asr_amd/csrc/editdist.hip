// Levenshtein distance between the greedy-decoded hypotheses and the references, on the device, for a whole batch:
// the WER / CER numerator and denominator of utils/asr_metrics.py:26-59,187-228 without the per-step D2H of the
// token ids and the Python loop over utterances.
//
//   CER mode (space_id < 0; the reference's file-path vocabularies, train.py:216-219): units = token ids.
//   WER mode (space_id >= 0): units = words = maximal runs of non-space tokens (str.split() semantics: leading /
//   trailing / repeated spaces produce no empty word); a word is compared through a 64-bit polynomial hash of its ids.
//
// One wave per utterance.  Row i of the DP (hypothesis unit i against every reference prefix j):
//   cur[j] = min(prev[j] + 1, prev[j-1] + (a_i != b_j), cur[j-1] + 1)
// The last term is a running dependency along j; with t[j] = min(prev[j] + 1, prev[j-1] + cost) it unrolls to
//   cur[j] = j + min_{k <= j} (t[k] - k),
// a prefix-min, so a row is: elementwise t, one wave prefix-min scan (6 DPP/shuffle steps), add j.  The reference row
// lives in registers (NJ cells per lane), `prev` never leaves the wave.
#include "common.h"
#include <algorithm>

namespace lasr {

static constexpr int kEdMaxUnits = 2048;          // units (tokens or words) per side
static constexpr int kEdNJ = kEdMaxUnits / 64;    // DP cells per lane

// sequence of DP units of one side -> LDS (u64), returns the count.  One lane walks the tokens (<= a few hundred).
__device__ __forceinline__ int ed_units(const int32_t* tok32, const int64_t* tok64, int n, int space_id, unsigned long long* s_u) {
  int cnt = 0;
  if (space_id < 0) {
    for (int i = 0; i < n && cnt < kEdMaxUnits; ++i) s_u[cnt++] = (unsigned long long)(tok32 ? (long long)tok32[i] : tok64[i]) + 1ull;
    return cnt;
  }
  unsigned long long h = 0ull;
  bool in_word = false;
  for (int i = 0; i < n; ++i) {
    const long long t = tok32 ? (long long)tok32[i] : tok64[i];
    if (t == space_id) {
      if (in_word && cnt < kEdMaxUnits) s_u[cnt++] = h;
      in_word = false; h = 0ull;
    } else {
      h = h * 1000003ull + (unsigned long long)(t + 1) * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull;   // order-sensitive
      in_word = true;
    }
  }
  if (in_word && cnt < kEdMaxUnits) s_u[cnt++] = h;
  return cnt;
}

__device__ __forceinline__ int wave_prefix_min_excl_carry(int v, int lane) {   // inclusive prefix-min over the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(v, d, 64);
    if (lane >= d) v = min(v, o);
  }
  return v;
}

// grid = B, block = 64.  hyp [B][ld_h] i32 with hyp_lens; ref [B][ld_r] i64 with ref_lens (i32).
// NJ = DP cells per lane: 64 * NJ >= the longer of (hypothesis, reference) pitch of the batch (a row costs NJ cell updates per lane
// whatever the actual length: 2048 cells per row for every batch cost 77 us at cfg2)
// dist[b] = Levenshtein(hyp units, ref units), ref_units[b] = number of reference units.
template <int NJ>
__global__ __launch_bounds__(64) void edit_distance_kernel(const int32_t* __restrict__ hyp, const int32_t* __restrict__ hyp_lens, int64_t ld_h,
                                                           const int64_t* __restrict__ ref, const int32_t* __restrict__ ref_lens, int64_t ld_r,
                                                           int space_id, int32_t* __restrict__ dist, int32_t* __restrict__ ref_units) {
  __shared__ unsigned long long s_a[kEdMaxUnits], s_b[kEdMaxUnits];
  __shared__ int s_n[2];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int nh = max(0, min(hyp_lens[b], (int)ld_h)), nr = max(0, min(ref_lens[b], (int)ld_r));
  if (space_id < 0) {             // CER: a unit is a token id, copied by the whole wave
    for (int i = lane; i < min(nh, kEdMaxUnits); i += 64) s_a[i] = (unsigned long long)(long long)hyp[(int64_t)b * ld_h + i] + 1ull;
    for (int i = lane; i < min(nr, kEdMaxUnits); i += 64) s_b[i] = (unsigned long long)ref[(int64_t)b * ld_r + i] + 1ull;
    if (lane == 0) { s_n[0] = min(nh, kEdMaxUnits); s_n[1] = min(nr, kEdMaxUnits); }
  } else if (lane == 0) {         // WER: one lane cuts the token stream into words (a few hundred tokens)
    s_n[0] = ed_units(hyp + (int64_t)b * ld_h, nullptr, nh, space_id, s_a);
    s_n[1] = ed_units(nullptr, ref + (int64_t)b * ld_r, nr, space_id, s_b);
  }
  __syncthreads();
  // The distance is symmetric: the LONGER sequence goes onto the lanes (the row cost does not depend on it), the loop runs over the
  // SHORTER one - an untrained model's greedy output is ~T' tokens against a 100-label reference: 501 dependent rows became 100.
  const int n_ref = s_n[1];
  const bool swap = s_n[0] > s_n[1];                       // wave-uniform
  const unsigned long long* s_rows = swap ? s_b : s_a;
  const unsigned long long* s_cols = swap ? s_a : s_b;
  const int na = swap ? s_n[1] : s_n[0], nb = swap ? s_n[0] : s_n[1];
  // lane owns column positions j = lane*NJ + q + 1 (q < NJ): contiguous per lane, so the in-lane part of the scan is serial
  unsigned long long bj[NJ];
  int prev[NJ];                // prev[q] = D[i-1][j]
#pragma unroll
  for (int q = 0; q < NJ; ++q) {
    const int j = lane * NJ + q + 1;
    bj[q] = j <= nb ? s_cols[j - 1] : 0ull;
    prev[q] = j;                  // D[0][j] = j
  }
  for (int i = 1; i <= na; ++i) {
    const unsigned long long ai = s_rows[i - 1];
    // D[i-1][j-1] for this lane's first cell comes from the previous lane's last cell (lane 0: D[i-1][0] = i-1)
    int left_prev = __shfl_up(prev[NJ - 1], 1, 64);
    if (lane == 0) left_prev = i - 1;
    int t[NJ];
    int run = 0x3fffffff;         // min over this lane's cells of t[k] - k
#pragma unroll
    for (int q = 0; q < NJ; ++q) {
      const int j = lane * NJ + q + 1;
      const int diag = q == 0 ? left_prev : prev[q - 1];
      const int v = min(prev[q] + 1, diag + (ai != bj[q] ? 1 : 0));
      run = min(run, v - j);
      t[q] = run;                 // in-lane prefix of (t - k)
    }
    // exclusive prefix-min over the lanes before this one, seeded with the column-0 term D[i][0] - 0 = i
    int incl = wave_prefix_min_excl_carry(run, lane);
    int before = __shfl_up(incl, 1, 64);
    if (lane == 0) before = 0x3fffffff;
    before = min(before, i);      // cur[0] = i contributes (i - 0) to every j
#pragma unroll
    for (int q = 0; q < NJ; ++q) {
      const int j = lane * NJ + q + 1;
      prev[q] = j + min(t[q], before);
    }
  }
  // D[na][nb]: nb = 0 -> na
  int res = na;
#pragma unroll
  for (int q = 0; q < NJ; ++q) {
    const int j = lane * NJ + q + 1;
    if (j == nb) res = prev[q];
  }
  const int owner = nb > 0 ? (nb - 1) / NJ : 0;
  res = __shfl(res, owner, 64);
  if (lane == 0) { dist[b] = res; ref_units[b] = n_ref; }
}

// totals[0] += sum dist, totals[1] += sum ref_units  (the Metric's `scores` / `words` states, utils/asr_metrics.py:114-115)
__global__ __launch_bounds__(64) void edit_totals_kernel(const int32_t* __restrict__ dist, const int32_t* __restrict__ ref_units, int64_t B,
                                                         long long* __restrict__ totals) {
  long long s = 0, w = 0;
  for (int64_t i = threadIdx.x; i < B; i += 64) { s += dist[i]; w += ref_units[i]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); w += __shfl_xor(w, o, 64); }
  if (threadIdx.x == 0) { totals[0] += s; totals[1] += w; }
}

// one training step's logged scalars folded into device accumulators (no D2H per step)
__global__ __launch_bounds__(64) void step_metrics_kernel(const float* __restrict__ loss, const int32_t* __restrict__ dist,
                                                          const int32_t* __restrict__ ref_units, int64_t B, double* __restrict__ acc) {
  long long s = 0, w = 0;
  for (int64_t i = threadIdx.x; i < B; i += 64) { s += dist[i]; w += ref_units[i]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); w += __shfl_xor(w, o, 64); }
  if (threadIdx.x == 0) {
    const double wer = w != 0 ? (double)s / (double)w : (double)INFINITY;    // utils/asr_metrics.py:59 (inf when no reference units)
    const double l = (double)loss[0];
    acc[0] += l; acc[1] += wer; acc[2] += 1.0; acc[3] = l; acc[4] = wer; acc[5] += (double)s; acc[6] += (double)w;
  }
}

}  // namespace lasr

using namespace lasr;

extern "C" int lasr_step_metrics(const float* loss, const int32_t* dist, const int32_t* ref_units, int64_t B, double* acc, void* stream) {
  LASR_CHECK_ARG(loss && dist && ref_units && acc && B > 0, "lasr_step_metrics: bad argument");
  hipLaunchKernelGGL(step_metrics_kernel, dim3(1), dim3(64), 0, as_stream(stream), loss, dist, ref_units, B, acc);
  LASR_LAUNCH_CHECK("step_metrics_kernel");
  return 0;
}

extern "C" int lasr_edit_distance_batch(const int32_t* hyp_tokens, const int32_t* hyp_lens, int64_t ld_hyp, const int64_t* ref_tokens,
                                        const int32_t* ref_lens, int64_t ld_ref, int64_t B, int space_id, int32_t* dist, int32_t* ref_units,
                                        int64_t* totals, void* stream) {
  LASR_CHECK_ARG(hyp_tokens && hyp_lens && ref_tokens && ref_lens && dist && ref_units, "lasr_edit_distance_batch: null pointer");
  LASR_CHECK_SHAPE(B > 0 && B < (1 << 20) && ld_hyp > 0 && ld_ref > 0, "lasr_edit_distance_batch: B=%lld", (long long)B);
  LASR_CHECK_SHAPE(ld_hyp <= kEdMaxUnits && ld_ref <= kEdMaxUnits, "lasr_edit_distance_batch: more than %d tokens per utterance",
                   kEdMaxUnits);
  hipStream_t st = as_stream(stream);
  const int64_t ld_cols = std::max(ld_hyp, ld_ref);        // either side may end up on the lanes
  if (ld_cols <= 128)
    hipLaunchKernelGGL(edit_distance_kernel<2>, dim3((unsigned)B), dim3(64), 0, st, hyp_tokens, hyp_lens, ld_hyp, ref_tokens, ref_lens, ld_ref,
                       space_id, dist, ref_units);
  else if (ld_cols <= 512)
    hipLaunchKernelGGL(edit_distance_kernel<8>, dim3((unsigned)B), dim3(64), 0, st, hyp_tokens, hyp_lens, ld_hyp, ref_tokens, ref_lens, ld_ref,
                       space_id, dist, ref_units);
  else
    hipLaunchKernelGGL(edit_distance_kernel<kEdNJ>, dim3((unsigned)B), dim3(64), 0, st, hyp_tokens, hyp_lens, ld_hyp, ref_tokens, ref_lens, ld_ref,
                       space_id, dist, ref_units);
  LASR_LAUNCH_CHECK("edit_distance_kernel");
  if (totals) {
    hipLaunchKernelGGL(edit_totals_kernel, dim3(1), dim3(64), 0, st, dist, ref_units, B, reinterpret_cast<long long*>(totals));
    LASR_LAUNCH_CHECK("edit_totals_kernel");
  }
  return 0;
}
