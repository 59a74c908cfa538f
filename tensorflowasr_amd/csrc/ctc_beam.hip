// CTC prefix beam search (CtcModel.recognize_beam, models/ctc/base_ctc.py:127-149 -> tf.nn.ctc_beam_search_decoder: a HOST op in
// the reference too).  Host code only: the caller hands over the logits of one batch in host memory; the search works in the
// log domain with the usual (blank-ending, label-ending) probability pair per prefix, no repeated-label merging beyond CTC's
// own collapse (merge_repeated = False in the v2 API), top path only.
//
// Reference quirk kept on purpose: tf.nn.ctc_beam_search_decoder always treats the LAST class as blank, while this code
// base's blank is 0 (base_ctc.py passes no blank index to the beam decoder) - so `blank_index` is a parameter and the Python
// host passes V-1 to reproduce recognize_beam, 0 for a self-consistent decoder.
#include "common.h"
#include <algorithm>
#include <cmath>
#include <map>
#include <vector>

namespace {

constexpr float NEG_INF = -INFINITY;
inline float lse2(float a, float b) {
  if (a == NEG_INF) return b;
  if (b == NEG_INF) return a;
  const float m = a > b ? a : b;
  return m + log1pf(expf(-fabsf(a - b)));
}
struct Beam { float pb, pnb; };

}  // namespace

extern "C" int tfasr_ctc_beam_search_host(const float* logits, const int32_t* logit_len, int B, int T, int V, int beam_width,
                                          int blank_index, int32_t* tokens, int32_t* tokens_len, float* log_prob) {
  if (!logits || !logit_len || !tokens || !tokens_len || B <= 0 || T <= 0 || V <= 1 || beam_width <= 0 || blank_index < 0 || blank_index >= V)
    return TFASR_STATUS_INVALID_VALUE;
  std::vector<float> lp(V);
  for (int b = 0; b < B; ++b) {
    const int Tb = std::min(std::max(logit_len[b], 0), T);
    std::map<std::vector<int32_t>, Beam> beams;
    beams[{}] = Beam{0.f, NEG_INF};
    for (int t = 0; t < Tb; ++t) {
      const float* x = logits + ((size_t)b * T + t) * V;
      float mx = x[0];
      for (int v = 1; v < V; ++v) mx = std::max(mx, x[v]);
      double se = 0.0;
      for (int v = 0; v < V; ++v) se += std::exp((double)x[v] - mx);
      const float lz = mx + (float)std::log(se);
      for (int v = 0; v < V; ++v) lp[v] = x[v] - lz;
      std::map<std::vector<int32_t>, Beam> next;
      auto slot = [&](const std::vector<int32_t>& k) -> Beam& {
        auto it = next.find(k);
        if (it == next.end()) it = next.emplace(k, Beam{NEG_INF, NEG_INF}).first;
        return it->second;
      };
      for (const auto& kv : beams) {
        const std::vector<int32_t>& pre = kv.first;
        const float pb = kv.second.pb, pnb = kv.second.pnb, ptot = lse2(pb, pnb);
        // stay on the prefix: emit blank, or repeat its last label
        Beam& same = slot(pre);
        same.pb = lse2(same.pb, ptot + lp[blank_index]);
        if (!pre.empty()) same.pnb = lse2(same.pnb, pnb + lp[pre.back()]);
        // extend by one label
        for (int c = 0; c < V; ++c) {
          if (c == blank_index) continue;
          const float add = (!pre.empty() && pre.back() == c) ? pb : ptot;  // a repeat needs a blank in between
          if (add == NEG_INF) continue;
          std::vector<int32_t> ext(pre);
          ext.push_back(c);
          Beam& e = slot(ext);
          e.pnb = lse2(e.pnb, add + lp[c]);
        }
      }
      // keep the beam_width most probable prefixes
      std::vector<std::pair<float, const std::vector<int32_t>*>> order;
      order.reserve(next.size());
      for (const auto& kv : next) order.emplace_back(lse2(kv.second.pb, kv.second.pnb), &kv.first);
      const size_t keep = std::min<size_t>(beam_width, order.size());
      std::partial_sort(order.begin(), order.begin() + keep, order.end(), [](const auto& a, const auto& c) {
        if (a.first != c.first) return a.first > c.first;
        return *a.second < *c.second;  // deterministic tie break
      });
      std::map<std::vector<int32_t>, Beam> pruned;
      for (size_t i = 0; i < keep; ++i) pruned.emplace(*order[i].second, next[*order[i].second]);
      beams.swap(pruned);
    }
    const std::vector<int32_t>* best = nullptr;
    float bestp = NEG_INF;
    for (const auto& kv : beams) {
      const float p = lse2(kv.second.pb, kv.second.pnb);
      if (!best || p > bestp || (p == bestp && kv.first < *best)) { best = &kv.first; bestp = p; }
    }
    int32_t* out = tokens + (size_t)b * T;
    const int n = best ? (int)best->size() : 0;
    for (int i = 0; i < T; ++i) out[i] = i < n ? (*best)[i] : 0;  // tf.sparse.to_dense default value 0
    tokens_len[b] = n;
    if (log_prob) log_prob[b] = bestp;
  }
  return TFASR_STATUS_SUCCESS;
}
