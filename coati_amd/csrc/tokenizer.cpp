// Trie tokenizer on the host: the reference tokenises every SMILES row in Python
// (tokenizers/trie.py:39-214 split, trie_tokenizer.py:48-78 pre_tokenize / tokenize_text) -- at > 20 k molecules/s per GPU
// that loop is the feed-rate bottleneck.  Same behaviour, C++:
//   * Trie::split = scan left to right; at every position not inside an earlier match take the LONGEST vocabulary word
//     starting there, otherwise the character joins the current unmatched chunk (leftmost-longest -- what the
//     reference's state machine with its look-ahead computes);
//   * pre_tokenize = split on the special-token trie first, then split the non-special chunks on the SMILES trie;
//   * tokenize = map every piece through the vocabulary; a piece that is not a key fails the row (KeyError there).
// Tries are byte tries over UTF-8: a vocabulary word starts on a lead byte, so byte-wise leftmost-longest matching never
// cuts inside a code point and equals the code-point-wise result.  Batch encoding fans rows out over std::thread.
#include <atomic>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace {

struct ByteTrie {
  struct Node {
    int next[256];
    int word;   // id of the word ending here (-2: a word without a vocabulary id); only valid when term
    bool term;
    Node() : word(-1), term(false) { for (int& n : next) n = -1; }
  };
  std::vector<Node> nodes;
  ByteTrie() : nodes(1) {}
  void add(const std::string& w, int id) {
    if (w.empty()) return;
    int cur = 0;
    for (unsigned char c : w) {
      if (nodes[cur].next[c] < 0) {
        nodes[cur].next[c] = (int)nodes.size();
        nodes.emplace_back();
      }
      cur = nodes[cur].next[c];
    }
    nodes[cur].word = id < 0 ? -2 : id;   // a duplicated word keeps the LAST id (the vocab dict: later index overwrites)
    nodes[cur].term = true;
  }
};

struct Piece { size_t begin, end; int id; };   // byte range; id = matched word of the trie used, -1 = other text

// byte offsets of the code points of text[begin, end) (+ the end offset): the reference iterates Python str characters
static void char_offsets(const char* text, size_t begin, size_t end, std::vector<size_t>& off) {
  off.clear();
  for (size_t i = begin; i < end; ++i)
    if (((unsigned char)text[i] & 0xC0) != 0x80) off.push_back(i);
  off.push_back(end);
}

// one trie step over the character [b, e): node index or -1   (`char in trie_pointer` / `trie_pointer[char]`)
static int step(const ByteTrie& t, int node, const char* text, size_t b, size_t e) {
  for (size_t i = b; i < e && node >= 0; ++i) node = t.nodes[node].next[(unsigned char)text[i]];
  return node;
}

// Trie.split (trie.py:39-191), the state machine itself, statement for statement -- including its look-ahead quirk (an
// earlier partial match that just FAILED on the current character is still extended from current + 1), so that texts
// with out-of-vocabulary characters split exactly like the reference.  For fully tokenisable text it reduces to
// leftmost-longest matching.  Returns false when cut_text would raise ("start > end").
bool split(const ByteTrie& t, const char* text, size_t begin, size_t end, std::vector<Piece>& out) {
  std::vector<size_t> off;
  char_offsets(text, begin, end, off);
  const int n = (int)off.size() - 1;
  auto terminal = [&](int node) { return t.nodes[node].term; };
  auto next = [&](int node, int ci) { return ci < n ? step(t, node, text, off[ci], off[ci + 1]) : -1; };
  std::vector<std::pair<int, int>> states;   // (start char index, trie node), insertion = increasing start
  std::vector<int> offsets{0};
  int skip = 0;
  for (int current = 0; current < n; ++current) {
    if (skip && current < skip) continue;
    std::vector<int> to_remove;
    bool reset = false;
    for (size_t si = 0; si < states.size(); ++si) {
      int start = states[si].first;
      const int ptr = states[si].second;
      if (terminal(ptr)) {
        int endc = current;
        for (size_t li = 0; li < states.size(); ++li) {
          const int lookstart = states[li].first;
          int lookptr = states[li].second;
          int la;
          if (lookstart > start) break;
          if (lookstart < start) { la = current + 1; endc = current + 1; }
          else { la = current; endc = current; }
          if (terminal(lookptr)) { start = lookstart; endc = la; skip = la; }
          int nx = next(lookptr, la);
          while (nx >= 0) {
            lookptr = nx;
            la += 1;
            if (terminal(lookptr)) { start = lookstart; endc = la; skip = la; }
            if (la == n) break;
            nx = next(lookptr, la);
          }
        }
        offsets.push_back(start);
        offsets.push_back(endc);
        reset = true;
        break;
      }
      const int nx = next(ptr, current);
      if (nx >= 0) states[si].second = nx;
      else to_remove.push_back(states[si].first);
    }
    if (reset) {
      states.clear();
    } else if (!to_remove.empty()) {
      std::vector<std::pair<int, int>> keep;
      for (auto& st : states) {
        bool rm = false;
        for (int r : to_remove) rm = rm || (r == st.first);
        if (!rm) keep.push_back(st);
      }
      states.swap(keep);
    }
    if (current >= skip) {
      const int nx = next(0, current);
      if (nx >= 0) states.push_back({current, nx});
    }
  }
  for (auto& st : states)
    if (terminal(st.second)) { offsets.push_back(st.first); offsets.push_back(n); break; }
  // cut_text (trie.py:193-214)
  offsets.push_back(n);
  int start = 0;
  for (int e : offsets) {
    if (start > e) return false;
    if (start == e) continue;
    Piece p{off[start], off[e], -1};
    // a piece that is exactly a vocabulary word carries its id
    int node = 0;
    for (size_t i = p.begin; i < p.end && node >= 0; ++i) node = t.nodes[node].next[(unsigned char)text[i]];
    if (node >= 0 && t.nodes[node].term) p.id = t.nodes[node].word;
    out.push_back(p);
    start = e;
  }
  return true;
}

}  // namespace

struct coati_tokenizer {
  ByteTrie special, smiles;
  int n_special = 0, n_token = 0;
};

namespace {
// returns the number of pieces, or -(1 + byte offset of the piece without a vocabulary id)
long long encode_one(const coati_tokenizer* tk, const char* text, size_t n, std::vector<int>& ids, std::vector<Piece>* pieces_out) {
  std::vector<Piece> top;
  ids.clear();
  if (pieces_out) pieces_out->clear();
  if (!split(tk->special, text, 0, n, top)) return -1;
  long long first_bad = 0;
  for (const Piece& p : top) {
    if (p.id != -1) {   // `T in self.special_tokens` (p.id may be -2: a special word whose text is not a vocab key)
      if (pieces_out) pieces_out->push_back(p);
      if (p.id < 0 && !first_bad) first_bad = -(long long)(1 + p.begin);
      ids.push_back(p.id);
      continue;
    }
    std::vector<Piece> sub;
    if (!split(tk->smiles, text, p.begin, p.end, sub)) return -1;
    for (const Piece& q : sub) {
      if (pieces_out) pieces_out->push_back(q);
      if (q.id < 0 && !first_bad) first_bad = -(long long)(1 + q.begin);
      ids.push_back(q.id);
    }
  }
  return first_bad ? first_bad : (long long)ids.size();
}
}  // namespace

extern "C" {

// special[i] has id special_ids[i] (default i), smiles[j] has id smiles_ids[j] (default n_special + j)
// (trie_tokenizer.py:22-24: keys = special_tokens + smiles_tokens, vocab = {key.strip(): index})
int coati_tokenizer_create(const char* const* special, const int32_t* special_ids, int n_special, const char* const* smiles,
                           const int32_t* smiles_ids, int n_smiles, coati_tokenizer** out) {
  COATI_CHECK_ARG(out && (special || n_special == 0) && (smiles || n_smiles == 0) && n_special >= 0 && n_smiles >= 0,
                  "tokenizer_create: bad arguments");
  coati_tokenizer* tk = new coati_tokenizer();
  tk->n_special = n_special;
  tk->n_token = n_special + n_smiles;
  for (int i = 0; i < n_special; ++i) tk->special.add(special[i], special_ids ? special_ids[i] : i);
  for (int j = 0; j < n_smiles; ++j) tk->smiles.add(smiles[j], smiles_ids ? smiles_ids[j] : n_special + j);
  *out = tk;
  return COATI_OK;
}

void coati_tokenizer_destroy(coati_tokenizer* tk) { delete tk; }

// ids_out[cap]; returns the number of tokens (may exceed cap: nothing past cap is written), or -(1 + byte offset) of the
// first piece that is not in the vocabulary (the reference raises KeyError there)
long long coati_tokenizer_encode(const coati_tokenizer* tk, const char* text, long long n_bytes, int32_t* ids_out, int cap) {
  if (!tk || !text || n_bytes < 0) return -1;
  std::vector<int> ids;
  const long long r = encode_one(tk, text, (size_t)n_bytes, ids, nullptr);
  if (r < 0) return r;
  for (long long i = 0; i < r && i < cap; ++i) ids_out[i] = ids[i];
  return r;
}

// pre_tokenize (trie_tokenizer.py:48-60): piece boundaries as byte offsets [begin, end) + id (< 0 = not in the
// vocabulary); returns the number of pieces, or -1 when the reference's cut_text would raise
long long coati_tokenizer_pieces(const coati_tokenizer* tk, const char* text, long long n_bytes, int64_t* begin, int64_t* end,
                                 int32_t* id, int cap) {
  if (!tk || !text || n_bytes < 0) return -1;
  std::vector<int> ids;
  std::vector<Piece> all;
  const long long r = encode_one(tk, text, (size_t)n_bytes, ids, &all);
  if (r == -1 && all.empty() && n_bytes > 0) return -1;
  for (size_t i = 0; i < all.size() && (long long)i < cap; ++i) { begin[i] = (int64_t)all[i].begin; end[i] = (int64_t)all[i].end; id[i] = all[i].id; }
  return (long long)all.size();
}

// rows[i] (NUL-terminated UTF-8) -> out[i, 0..n_seq) int64, zero ([PAD]) padded; len[i] = token count, or -1 when a piece
// is not in the vocabulary, or -2 when the row is longer than n_seq (the row is then left all-zero).  n_threads <= 0: auto.
int coati_tokenizer_encode_batch(const coati_tokenizer* tk, const char* const* rows, int n_rows, int n_seq, int64_t* out,
                                 int32_t* len, int n_threads) {
  COATI_CHECK_ARG(tk && rows && out && len && n_rows >= 0 && n_seq > 0, "tokenizer_encode_batch: bad arguments");
  if (n_threads <= 0) {
    n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads <= 0) n_threads = 1;
    if (n_threads > 32) n_threads = 32;
  }
  if (n_threads > n_rows / 64 + 1) n_threads = n_rows / 64 + 1;
  std::atomic<int> next(0);
  auto work = [&]() {
    std::vector<int> ids;
    for (;;) {
      const int i0 = next.fetch_add(64);
      if (i0 >= n_rows) break;
      const int i1 = i0 + 64 < n_rows ? i0 + 64 : n_rows;
      for (int i = i0; i < i1; ++i) {
        int64_t* dst = out + (size_t)i * n_seq;
        std::memset(dst, 0, sizeof(int64_t) * n_seq);
        const long long r = encode_one(tk, rows[i], std::strlen(rows[i]), ids, nullptr);
        if (r < 0) { len[i] = -1; continue; }
        if (r > n_seq) { len[i] = -2; continue; }
        for (long long t = 0; t < r; ++t) dst[t] = ids[t];
        len[i] = (int32_t)r;
      }
    }
  };
  if (n_threads == 1) {
    work();
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) th.emplace_back(work);
    for (auto& t : th) t.join();
  }
  return COATI_OK;
}

}  // extern "C"
