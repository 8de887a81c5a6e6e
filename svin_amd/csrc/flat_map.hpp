// svin_amd host core: a flat hash map for the per-observation bookkeeping (no HIP, no allocation per entry).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace svin {

// open-addressing map uint64 -> uint64 (linear probing, backward-shift deletion): residual id -> landmark id and landmark
// id -> handle are looked up once per addObservation / removeObservation -- a node-based std::unordered_map spends more time
// in malloc / free than the rest of the call
class FlatMap64 {
 public:
  static constexpr uint64_t kEmpty = UINT64_MAX;   // (values are handles and node addresses: never this)
  FlatMap64() { rehash(1024); }
  size_t size() const { return n_; }
  bool find(uint64_t key, uint64_t* val) const {
    for (size_t i = slot(key);; i = (i + 1) & mask_) {
      if (e_[i].val == kEmpty) return false;
      if (e_[i].key == key) { if (val) *val = e_[i].val; return true; }
    }
  }
  bool count(uint64_t key) const { return find(key, nullptr); }
  void set(uint64_t key, uint64_t val) {
    if ((n_ + 1) * 2 > mask_ + 1) rehash(2 * (mask_ + 1));
    for (size_t i = slot(key);; i = (i + 1) & mask_) {
      if (e_[i].val == kEmpty) { e_[i].key = key; e_[i].val = val; ++n_; return; }
      if (e_[i].key == key) { e_[i].val = val; return; }
    }
  }
  bool erase(uint64_t key) {
    size_t i = slot(key);
    for (;; i = (i + 1) & mask_) {
      if (e_[i].val == kEmpty) return false;
      if (e_[i].key == key) break;
    }
    for (size_t j = (i + 1) & mask_;; j = (j + 1) & mask_) {   // close the gap: move back every entry that probes through it
      if (e_[j].val == kEmpty) break;
      const size_t home = slot(e_[j].key);
      if (((j - home) & mask_) >= ((j - i) & mask_)) { e_[i] = e_[j]; i = j; }
    }
    e_[i].val = kEmpty;
    --n_;
    return true;
  }

 private:
  struct Entry { uint64_t key, val; };
  size_t slot(uint64_t k) const { return (size_t)((k * 0x9E3779B97F4A7C15ull) >> 20) & mask_; }
  void rehash(size_t cap) {
    std::vector<Entry> old = std::move(e_);
    e_.assign(cap, Entry{0, kEmpty});
    mask_ = cap - 1; n_ = 0;
    for (const Entry& en : old) if (en.val != kEmpty) set(en.key, en.val);
  }
  std::vector<Entry> e_;
  size_t mask_ = 0, n_ = 0;
};

}  // namespace svin
