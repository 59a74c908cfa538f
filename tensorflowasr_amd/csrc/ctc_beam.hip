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
#include <unordered_map>
#include <vector>

namespace {

constexpr float NEG_INF = -INFINITY;
inline float lse2(float a, float b) {
  if (a == NEG_INF) return b;
  if (b == NEG_INF) return a;
  const float m = a > b ? a : b;
  return m + log1pf(expf(-fabsf(a - b)));
}

// Prefixes live in a trie (node = parent + last label), so extending a prefix costs O(1) instead of a vector copy + an ordered-map
// insertion keyed by the whole label sequence (the first version spent 79 s on 32 x 250 frames x 1000 classes at width 10).
struct Node { int parent, label, depth; };
struct Beam { int node; float pb, pnb; };
struct Cand { float total; int src; int label; };  // label >= 0: beam `src` extended by `label`; label < 0: beam `src` itself

}  // namespace

extern "C" int tfasr_ctc_beam_search_host(const float* logits, const int32_t* logit_len, int B, int T, int V, int beam_width,
                                          int blank_index, int32_t* tokens, int32_t* tokens_len, float* log_prob) {
  if (!logits || !logit_len || !tokens || !tokens_len || B <= 0 || T <= 0 || V <= 1 || beam_width <= 0 || blank_index < 0 || blank_index >= V)
    return TFASR_STATUS_INVALID_VALUE;
  std::vector<float> lp(V);
  std::vector<Node> trie;
  std::vector<Beam> beams, next;
  std::vector<float> ext;        // [nb][V] label-ending probability of beam i extended by label c (incl. merged "stay" mass)
  std::vector<float> ext_pb;     // [nb][V] blank-ending probability of that same prefix when it is ALSO a current beam
  std::vector<Cand> cand;
  std::vector<int32_t> pa, pc;
  // (parent node, label) -> node, for the whole utterance: a label sequence has exactly ONE node, also when a prefix drops out of the
  // beam and is re-created later from its parent while one of its own extensions stayed (their probability mass must merge, as in
  // TensorFlow's beam-entry children map)
  std::unordered_map<uint64_t, int> child;
  auto prefix_of = [&](int node, std::vector<int32_t>& out) {
    out.clear();
    for (int n = node; n > 0; n = trie[n].parent) out.push_back(trie[n].label);
    std::reverse(out.begin(), out.end());
  };
  for (int b = 0; b < B; ++b) {
    const int Tb = std::min(std::max(logit_len[b], 0), T);
    trie.clear();
    child.clear();
    trie.push_back(Node{-1, -1, 0});  // node 0 = the empty prefix
    beams.assign(1, Beam{0, 0.f, NEG_INF});
    for (int t = 0; t < Tb; ++t) {
      const float* x = logits + ((size_t)b * T + t) * V;
      float mx = x[0];
      for (int v = 1; v < V; ++v) mx = std::max(mx, x[v]);
      double se = 0.0;
      for (int v = 0; v < V; ++v) se += std::exp((double)x[v] - mx);
      const float lz = mx + (float)std::log(se);
      for (int v = 0; v < V; ++v) lp[v] = x[v] - lz;
      const int nb = (int)beams.size();
      ext.assign((size_t)nb * V, NEG_INF);
      ext_pb.assign((size_t)nb * V, NEG_INF);
      // "stay" candidates: prefix unchanged (emit blank, or repeat its last label).  If the prefix's parent is itself a current beam,
      // the same prefix is also reachable as (parent extended by its last label): both contributions belong to ONE candidate.
      std::vector<int> merged_into(nb, -1);
      for (int i = 0; i < nb; ++i) {
        const Node& nd = trie[beams[i].node];
        if (nd.parent < 0) continue;
        for (int j = 0; j < nb; ++j)
          if (beams[j].node == nd.parent) { merged_into[i] = j; break; }
      }
      cand.clear();
      // extensions of every beam by one label
      for (int i = 0; i < nb; ++i) {
        const float pb = beams[i].pb, ptot = lse2(pb, beams[i].pnb);
        const Node& nd = trie[beams[i].node];
        float* row = ext.data() + (size_t)i * V;
        for (int c = 0; c < V; ++c) {
          if (c == blank_index) continue;
          const float add = (nd.label == c && nd.parent >= 0) ? pb : ptot;  // a repeat needs a blank in between
          if (add != NEG_INF) row[c] = add + lp[c];
        }
      }
      // staying on a prefix (emit blank, or repeat its last label): merged into (parent, last label) when the parent is a beam too
      for (int i = 0; i < nb; ++i) {
        const Node& nd = trie[beams[i].node];
        const float s_pb = lse2(beams[i].pb, beams[i].pnb) + lp[blank_index];
        const float s_pnb = nd.parent >= 0 ? beams[i].pnb + lp[nd.label] : NEG_INF;
        if (merged_into[i] >= 0) {
          const size_t k = (size_t)merged_into[i] * V + nd.label;
          ext_pb[k] = s_pb;
          ext[k] = lse2(ext[k], s_pnb);
        } else {
          cand.push_back(Cand{lse2(s_pb, s_pnb), i, -1});
        }
      }
      for (int i = 0; i < nb; ++i)
        for (int c = 0; c < V; ++c) {
          const size_t k = (size_t)i * V + c;
          if (ext[k] != NEG_INF || ext_pb[k] != NEG_INF) cand.push_back(Cand{lse2(ext_pb[k], ext[k]), i, c});
        }
      const size_t keep = std::min<size_t>(beam_width, cand.size());
      auto better = [&](const Cand& a, const Cand& c) {
        if (a.total != c.total) return a.total > c.total;
        // exact tie: order by the label sequences (what the reference-style ordered container did)
        prefix_of(beams[a.src].node, pa);
        if (a.label >= 0) pa.push_back(a.label);
        prefix_of(beams[c.src].node, pc);
        if (c.label >= 0) pc.push_back(c.label);
        return pa < pc;
      };
      std::partial_sort(cand.begin(), cand.begin() + keep, cand.end(), better);
      next.clear();
      for (size_t q = 0; q < keep; ++q) {
        const Cand& cd = cand[q];
        if (cd.label < 0) {
          const Beam& bm = beams[cd.src];
          const Node& nd = trie[bm.node];
          const float ptot = lse2(bm.pb, bm.pnb);
          next.push_back(Beam{bm.node, ptot + lp[blank_index], nd.parent >= 0 ? bm.pnb + lp[nd.label] : NEG_INF});
        } else {
          const size_t k = (size_t)cd.src * V + cd.label;
          // the extended prefix may already be a node (it is a beam now, or was one earlier): reuse it
          const int par = beams[cd.src].node;
          const uint64_t key = (uint64_t)par * (uint64_t)V + (uint64_t)cd.label;
          int node;
          auto it = child.find(key);
          if (it != child.end()) {
            node = it->second;
          } else {
            trie.push_back(Node{par, cd.label, trie[par].depth + 1});
            node = (int)trie.size() - 1;
            child.emplace(key, node);
          }
          next.push_back(Beam{node, ext_pb[k], ext[k]});
        }
      }
      beams.swap(next);
    }
    int best = -1;
    float bestp = NEG_INF;
    for (int i = 0; i < (int)beams.size(); ++i) {
      const float p = lse2(beams[i].pb, beams[i].pnb);
      bool take = best < 0 || p > bestp;
      if (!take && p == bestp) {
        prefix_of(beams[i].node, pa);
        prefix_of(beams[best].node, pc);
        take = pa < pc;
      }
      if (take) { best = i; bestp = p; }
    }
    int32_t* out = tokens + (size_t)b * T;
    std::vector<int32_t> seq;
    if (best >= 0) prefix_of(beams[best].node, seq);
    const int n = (int)seq.size();
    for (int i = 0; i < T; ++i) out[i] = i < n ? seq[i] : 0;  // tf.sparse.to_dense default value 0
    tokens_len[b] = n;
    if (log_prob) log_prob[b] = bestp;
  }
  return TFASR_STATUS_SUCCESS;
}
